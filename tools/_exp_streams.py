import time, sys, os
import torch
import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = 1 << log2n
eng0 = gs.Engine(0, 0)
shapes = [(0, N // 2), (1, N // 4), (2, N // 4)]
wls = [Workload(eng0, ty=t, N=n, m=4, n=4, seed=20241222) for t, n in shapes]
crs = wls[0].crs
def step_on(engs):
    for e, w in zip(engs, wls):
        e.prove_batch_dev(w.ty, w.N, w.m, w.n, w.X, w.Y, w.A, w.B, w.Gamma, w.R, w.S, w.T, w.xcoms, w.ycoms, w.pi, w.theta)
    for e, w in zip(engs, wls):
        e.verify_batch_dev(w.ty, w.N, w.m, w.n, w.A, w.B, w.Gamma, w.target, w.xcoms, w.ycoms, w.pi, w.theta, w.ok)
    for e in set(engs):
        e.sync()
def timeit(engs, label, reps=5):
    step_on(engs); step_on(engs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step_on(engs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(label, "%.1f ms" % (dt * 1e3), "%.0f /s" % (N / dt), flush=True)
timeit([eng0] * 3, "one context, default stream")
streams = [torch.cuda.Stream() for _ in range(3)]
engs = []
for s in streams:
    e = gs.Engine(0, 0)
    e.set_crs(crs)
    e.set_stream(s.cuda_stream)
    engs.append(e)
timeit(engs, "three contexts, three streams")
timeit([engs[0]] * 3, "one context on a non-default stream")
timeit([eng0] * 3, "one context, default stream (again)")
