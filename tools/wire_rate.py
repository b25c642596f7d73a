"""Throughput of the wire-format codecs (host pointers in, host pointers out; H2D/D2H included):
python tools/wire_rate.py  -> points / s for G1, G2 compressed and uncompressed, with and without validation."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload

N = 1 << 14
for curve in (0, 1):
    eng = gs.Engine(curve, 0)
    wl = Workload(eng, N=N // 4, m=4, n=4)
    pts = {"g1": wl.X.cpu().numpy().reshape(N, -1), "g2": wl.Y.cpu().numpy().reshape(N, -1)}
    for kind in ("g1", "g2"):
        for comp in (True, False):
            enc = eng.wire_encode(kind, pts[kind], comp)
            t0 = time.perf_counter()
            enc = eng.wire_encode(kind, pts[kind], comp)
            te = time.perf_counter() - t0
            for val in (False, True):
                eng.wire_decode(kind, enc, comp, val)
                t0 = time.perf_counter()
                out, ok = eng.wire_decode(kind, enc, comp, val)
                td = time.perf_counter() - t0
                assert ok.all() and (out == pts[kind]).all()
                print("curve %d %s %-12s encode %8.0f k/s   decode(validate=%d) %8.0f k/s" % (
                    curve, kind, "compressed" if comp else "uncompressed", N / te / 1e3, val, N / td / 1e3))
    eng.close()
