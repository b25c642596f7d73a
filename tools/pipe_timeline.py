"""One host-pointer prove + verify of 2^16 PPEs (after two warm-up calls) for a rocprofv3 timeline:
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o tl -- python3 tools/pipe_timeline.py [staged|pinned] [log2 N]
tools/pipe_timeline_report.py then prints kernels and copies of the last call pair on one time axis."""
import sys

import numpy as np
import torch

import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload

N = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
eng = gs.Engine(0, 0)
wl = Workload(eng, N=N, corrupt_every=0)
h = lambda t: t.cpu().numpy()
X, Y, A, B, G, R, S, T, tgt = map(h, (wl.X, wl.Y, wl.A, wl.B, wl.Gamma, wl.R, wl.S, wl.T, wl.target))
wl.prove()
eng.sync()
keep = {k: np.zeros_like(h(getattr(wl, k))) for k in ("xcoms", "ycoms", "pi", "theta")}
okbuf = np.zeros(N, dtype=np.uint8)
if len(sys.argv) > 1 and sys.argv[1] == "pinned":
    for a in [X, Y, A, B, G, R, S, T, tgt, okbuf] + list(keep.values()):
        eng.host_register(a)
for rep in range(3):
    eng.prove_batch(0, N, 4, 4, X, Y, A, B, G, R, S, T, out=keep)
    eng.verify_batch(0, N, 4, 4, A, B, G, tgt, keep["xcoms"], keep["ycoms"], keep["pi"], keep["theta"], ok=okbuf)
assert okbuf.all()
