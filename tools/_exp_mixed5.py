import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload
N = 1 << int(sys.argv[1])
eng = gs.Engine(0, 0)
wls = [Workload(eng, ty=t, N=n, m=4, n=4, seed=20241222) for t, n in [(0, N // 2), (1, N // 4), (2, N // 4)]]
pparts = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, X=w.X, Y=w.Y, A=w.A, B=w.B, Gamma=w.Gamma, R=w.R, S=w.S, T=w.T,
               xcoms=w.xcoms, ycoms=w.ycoms, pi=w.pi, theta=w.theta) for w in wls]
vparts = [dict(ty=w.ty, N=w.N, m=w.m, n=w.n, A=w.A, B=w.B, Gamma=w.Gamma, target=w.target, xcoms=w.xcoms,
               ycoms=w.ycoms, pi=w.pi, theta=w.theta, ok=w.ok) for w in wls]
for rep in range(3):
    eng.prove_mixed_dev(pparts); eng.verify_mixed_dev(vparts); eng.sync()
