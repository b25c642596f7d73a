import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import bench
class A: pass
b = bench.Bench(A(), None, 0, 1, 0, "cuda:0", "cuda:0")
for s in sys.argv[1:]:
    kw = dict(p12=dict(log2n=12), m12=dict(log2n=12, mixed=True), p16=dict(log2n=16), m16=dict(log2n=16, mixed=True), p14=dict(log2n=14), m14=dict(log2n=14, mixed=True))[s]
    r = b.run(steps=2, warmup=1, seed_off=3, roofline=True, **kw)
    print(s, round(r["value"]), round(r["ms_per_step"], 1), {k: v for k, v in r["roofline"]["kernels_ms"].items() if v > 3}, flush=True)
    print("   mem allocated by torch %.1f GB; free/total %s" % (torch.cuda.memory_allocated() / 1e9, [round(x / 1e9, 1) for x in torch.cuda.mem_get_info()]), flush=True)
