"""PCIe-inclusive rate of the host-pointer entry points (gs_prove_batch / gs_verify_batch):
inputs start in host memory, results end in host memory.  Reported in DESIGN.md only;
bench.py's `value` is the device-resident rate."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = gs.Engine(0, 0)
wl = Workload(eng, N=N)
h = lambda t: t.cpu().numpy()
X, Y, A, B, G, R, S, T, tgt = map(h, (wl.X, wl.Y, wl.A, wl.B, wl.Gamma, wl.R, wl.S, wl.T, wl.target))
for it in range(3):
    t0 = time.perf_counter()
    out = eng.prove_batch(0, N, 4, 4, X, Y, A, B, G, R, S, T)
    ok = eng.verify_batch(0, N, 4, 4, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], out["theta"])
    dt = time.perf_counter() - t0
    assert ok.all()
    print("host path: %d units in %.1f ms = %.0f proofs+verifies/s (H2D + D2H + staging allocations included)" % (N, dt * 1e3, N / dt))
