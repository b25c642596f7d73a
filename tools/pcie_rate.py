"""PCIe copy rates of this box as the runtime delivers them (the host-pointer path's floor): page-locked memory
(hipHostMalloc via torch's pinned allocator, and a hipHostRegister'ed numpy array through the library's own pipeline is
covered by tools/pipe_trace.py) to the device and back, one stream and two streams at once.
    python tools/pcie_rate.py [MB]"""
import sys
import time

import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mb << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def rate(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def one_h2d():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)


def two_h2d():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        d2.copy_(h2, non_blocking=True)


def one_d2h():
    with torch.cuda.stream(s1):
        h.copy_(d, non_blocking=True)


def both_dirs():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)


for name, fn, total in (("H2D one stream", one_h2d, n), ("H2D two streams", two_h2d, 2 * n), ("D2H one stream", one_d2h, n),
                        ("H2D + D2H at once", both_dirs, 2 * n)):
    t = rate(fn)
    print("%-20s %6d MB in %7.2f ms = %6.1f GB/s" % (name, total >> 20, t * 1e3, total / t / 1e9), flush=True)
for small in (1, 4, 16, 64):
    hs, ds = h[: small << 20], d[: small << 20]

    def f():
        with torch.cuda.stream(s1):
            ds.copy_(hs, non_blocking=True)

    t = rate(f, 20)
    print("H2D %3d MB piece: %7.3f ms = %6.1f GB/s" % (small, t * 1e3, (small << 20) / t / 1e9), flush=True)
