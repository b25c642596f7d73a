"""Mixed batch (BASELINE configs[2]: 50 % PPE, 25 % MSMEG1, 25 % MSMEG2, m = n = 4) through gs_prove_mixed_dev +
gs_verify_mixed_dev: merged segmented launches against the parts one after the other, next to a PPE-only batch of the
same size on the same box.      python tools/mixed_rate.py [log2 N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload

for log2n in [int(a) for a in sys.argv[1:]] or [12]:
    N = 1 << log2n
    eng = gs.Engine(0, 0)
    wls = [Workload(eng, ty=t, N=n, m=4, n=4, seed=20241222) for t, n in [(0, N // 2), (1, N // 4), (2, N // 4)]]
    pp = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, X=w.X, Y=w.Y, A=w.A, B=w.B, Gamma=w.Gamma, R=w.R, S=w.S, T=w.T,
               xcoms=w.xcoms, ycoms=w.ycoms, pi=w.pi, theta=w.theta) for w in wls]
    vp = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, A=w.A, B=w.B, Gamma=w.Gamma, target=w.target, xcoms=w.xcoms,
               ycoms=w.ycoms, pi=w.pi, theta=w.theta, ok=w.ok) for w in wls]
    steps = max(3, min(20, (1 << 17) // N))

    def timed(fn):
        fn(); fn()
        eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        eng.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def mixed():
        eng.prove_mixed_dev(pp)
        eng.verify_mixed_dev(vp)

    out = {}
    for merge in (0, 1, -1):
        eng.set_option("mixed_merge", merge)
        out[merge] = timed(mixed)
        assert all(w.ok.cpu().numpy().all() for w in wls)
    ppe = Workload(eng, ty=0, N=N, m=4, n=4, seed=20241222)
    t_ppe = timed(ppe.step)
    print("2^%d mixed 50/25/25: parts in sequence %.1f ms (%.0f /s), merged launches %.1f ms (%.0f /s), planned %.1f ms; "
          "PPE-only 2^%d %.1f ms (%.0f /s)" % (log2n, out[0], N / out[0] * 1e3, out[1], N / out[1] * 1e3, out[-1], log2n,
                                              t_ppe, N / t_ppe * 1e3), flush=True)
    eng.close()
