"""PCIe-inclusive rate of the host-pointer entry points (gs_prove_batch / gs_verify_batch): inputs start in host memory,
results end in host memory, next to the device-resident rate of the same batch on the same box.  Three callers:
  fresh   pageable inputs, output arrays allocated per call (first touch of their pages happens inside the call)
  reused  pageable inputs, output arrays the caller keeps across calls
  pinned  inputs and outputs registered once with gs_host_register (DMA straight from / to them, no staging copy)
    python tools/host_path_rate.py [log2 N] [steps]          (GS_COPY_THREADS=k python ... for the worker sweep)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 1 << log2n
eng = gs.Engine(0, 0)
wl = Workload(eng, N=N, corrupt_every=0)
h = lambda t: t.cpu().numpy()
X, Y, A, B, G, R, S, T, tgt = map(h, (wl.X, wl.Y, wl.A, wl.B, wl.Gamma, wl.R, wl.S, wl.T, wl.target))


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


def dev_prove():
    wl.prove()
    eng.sync()


def dev_verify():
    wl.verify()
    eng.sync()


tp_d, _ = timed(dev_prove, steps)
tv_d, _ = timed(dev_verify, steps)
inb = sum(a.nbytes for a in (X, Y, A, B, G, R, S, T))
want_pi = h(wl.pi)


def report(tag, tp_h, tv_h, outb):
    vin = A.nbytes + B.nbytes + G.nbytes + tgt.nbytes + outb
    print("2^%d PPE %-6s GS_COPY_THREADS=%s: prove dev %.1f host %.1f ms (+%.1f: %.0f MB in, %.0f MB out); verify dev %.1f "
          "host %.1f ms (+%.1f: %.0f MB in); host/dev rate ratio %.3f; host %.0f /s" % (
              log2n, tag, os.environ.get("GS_COPY_THREADS", "4"), tp_d, tp_h, tp_h - tp_d, inb / 1e6, outb / 1e6, tv_d,
              tv_h, tv_h - tv_d, vin / 1e6, (tp_d + tv_d) / (tp_h + tv_h), N / (tp_h + tv_h) * 1e3), flush=True)


# fresh output arrays per call
tp_h, out = timed(lambda: eng.prove_batch(0, N, 4, 4, X, Y, A, B, G, R, S, T), steps)
tv_h, ok = timed(lambda: eng.verify_batch(0, N, 4, 4, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], out["theta"]), steps)
assert ok.all() and (out["pi"] == want_pi).all()
outb = sum(v.nbytes for v in out.values())
report("fresh", tp_h, tv_h, outb)

# the caller keeps its output arrays
keep = {k: np.zeros_like(v) for k, v in out.items()}
okbuf = np.zeros(N, dtype=np.uint8)
tp_h, out2 = timed(lambda: eng.prove_batch(0, N, 4, 4, X, Y, A, B, G, R, S, T, out=keep), steps)
tv_h, ok = timed(lambda: eng.verify_batch(0, N, 4, 4, A, B, G, tgt, keep["xcoms"], keep["ycoms"], keep["pi"], keep["theta"],
                                          ok=okbuf), steps)
assert ok.all() and (keep["pi"] == want_pi).all()
report("reused", tp_h, tv_h, outb)

# everything page-locked once: buffers on mappings of their own (Engine.host_buffer), never heap memory
def fresh(a):
    b = eng.host_buffer(a.nbytes)
    b[:] = a.reshape(-1).view(np.uint8)
    return b


X, Y, A, B, G, R, S, T, tgt = [fresh(a) for a in (X, Y, A, B, G, R, S, T, tgt)]
keep = {k: eng.host_buffer(v.nbytes) for k, v in keep.items()}
okbuf = eng.host_buffer(N)
regs = [X, Y, A, B, G, R, S, T, tgt, okbuf] + list(keep.values())
t0 = time.perf_counter()
for a in regs:
    eng.host_register(a)
t_reg = (time.perf_counter() - t0) * 1e3
for v in keep.values():
    v[:] = 0
tp_h, out2 = timed(lambda: eng.prove_batch(0, N, 4, 4, X, Y, A, B, G, R, S, T, out=keep), steps)
tv_h, ok = timed(lambda: eng.verify_batch(0, N, 4, 4, A, B, G, tgt, keep["xcoms"], keep["ycoms"], keep["pi"], keep["theta"],
                                          ok=okbuf), steps)
assert ok.all() and (keep["pi"] == want_pi).all() and (keep["theta"] == h(wl.theta)).all()
report("pinned", tp_h, tv_h, outb)
print("  (registering the %d arrays, %.0f MB: %.1f ms once)" % (len(regs), sum(a.nbytes for a in regs) / 1e6, t_reg))
for a in regs:
    eng.host_unregister(a)
