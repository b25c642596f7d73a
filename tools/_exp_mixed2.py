import time, sys, os
import torch
import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload
eng = gs.Engine(0, 0)
big = Workload(eng, ty=0, N=1 << 16, m=4, n=4, seed=20241221)
big.step(); eng.sync()
t0 = time.perf_counter(); big.step(); eng.sync(); print("2^16 step %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
del big; torch.cuda.empty_cache()
N = 1 << 12
wls = [Workload(eng, ty=t, N=n, m=4, n=4, seed=20241222) for t, n in [(0, N // 2), (1, N // 4), (2, N // 4)]]
pparts = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, X=w.X, Y=w.Y, A=w.A, B=w.B, Gamma=w.Gamma, R=w.R, S=w.S, T=w.T,
               xcoms=w.xcoms, ycoms=w.ycoms, pi=w.pi, theta=w.theta) for w in wls]
vparts = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, A=w.A, B=w.B, Gamma=w.Gamma, target=w.target, xcoms=w.xcoms,
               ycoms=w.ycoms, pi=w.pi, theta=w.theta, ok=w.ok) for w in wls]
for rep in range(4):
    t0 = time.perf_counter(); eng.prove_mixed_dev(pparts); t1 = time.perf_counter()
    eng.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.verify_mixed_dev(vparts); t3 = time.perf_counter()
    eng.sync(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("mixed: prove enqueue %.1f ms, drain %.1f ms, verify enqueue %.1f ms, drain %.1f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    for w in wls: w.prove()
    t1 = time.perf_counter(); eng.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
    for w in wls: w.verify()
    t3 = time.perf_counter(); eng.sync(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("serial: prove enqueue %.1f ms, drain %.1f ms, verify enqueue %.1f ms, drain %.1f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)), flush=True)
