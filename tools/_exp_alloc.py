import time, torch
def bw(x, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s = x.sum()
    torch.cuda.synchronize()
    return x.numel() * 4 * reps / (time.perf_counter() - t0) / 1e9
def gather_bw(x, idx, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s = x[idx].sum()
    torch.cuda.synchronize()
    return idx.numel() * reps / (time.perf_counter() - t0) / 1e9
bufs = []
g = torch.Generator(device="cuda"); g.manual_seed(1)
for i in range(24):
    x = torch.ones(1 << 30, dtype=torch.int32, device="cuda")   # 4 GB
    idx = torch.randint(0, x.numel(), (1 << 24,), device="cuda", generator=g)
    bufs.append(x)
    print("buffer %2d (after %3d GB): stream %.0f GB/s, random gather %.2f G elem/s" % (i, 4 * i, bw(x), gather_bw(x, idx)), flush=True)
# churn: free every other, allocate odd sizes
for i in range(0, 24, 2): bufs[i] = None
torch.cuda.empty_cache()
for i in range(6):
    x = torch.ones((1 << 30) + (1 << 28), dtype=torch.int32, device="cuda")   # 5 GB into 4 GB holes
    idx = torch.randint(0, x.numel(), (1 << 24,), device="cuda", generator=g)
    bufs.append(x)
    print("after churn %d: stream %.0f GB/s, random gather %.2f G elem/s" % (i, bw(x), gather_bw(x, idx)), flush=True)
