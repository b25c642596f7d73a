"""The reference's large bench statement (benches/bench.rs:451-498, 531-578: one PPE with m = n = 334) on the GPU:
latency of commit_and_prove and of verify for N = 1 and N = 16 such equations, and the kernels that carry it.
Reported in DESIGN.md; not part of bench.py's headline line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 334
out = {}
# A/B of the verifier's Gamma^T c: Straus lanes with their own tables (var_tab = 0) against window
# tables of the 2 m commitment components shared by all 2 n outputs (var_tab = 1: k_tab_build + k_var_tab8)
for N, tab in ((1, 0), (1, 1), (16, 0), (16, 1)):
    eng = gs.Engine(0, 0)
    eng.set_option("var_tab", tab)
    wl = Workload(eng, ty=0, N=N, m=m, n=n, seed=334 + N, corrupt_every=0)
    for _ in range(2):
        wl.prove()
        wl.verify()
    eng.sync()
    tp = tv = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        wl.prove()
        eng.sync()
        t1 = time.perf_counter()
        wl.verify()
        eng.sync()
        t2 = time.perf_counter()
        tp, tv = min(tp, t1 - t0), min(tv, t2 - t1)
    assert wl.ok.cpu().numpy().all()
    eng.prof_enable(True)
    eng.prof_reset()
    wl.prove()
    wl.verify()
    eng.sync()
    prof = {p[0]: round(p[1], 3) for p in eng.prof_get()}
    eng.prof_enable(False)
    out["N=%d %s" % (N, "shared base tables (k_var_tab8)" if tab else "Straus lanes")] = {
        "prove_ms": tp * 1e3, "verify_ms": tv * 1e3, "gamma_msm_ms": sum(v for k, v in prof.items() if ".vg1" in k),
        "kernels_ms": prof}
    eng.close()
print(json.dumps({"shape": "PPE %dx%d BLS12-381" % (m, n), **out}))
