import os, sys, time
import numpy as np, torch
import groth_sahai_rs_amd as gs
from groth_sahai_rs_amd.workload import Workload
N = 1 << 16
eng = gs.Engine(0, 0)
wl = Workload(eng, N=N, corrupt_every=0)
h = lambda t: t.cpu().numpy()
X, Y, A, B, G, R, S, T, tgt = map(h, (wl.X, wl.Y, wl.A, wl.B, wl.Gamma, wl.R, wl.S, wl.T, wl.target))
wl.prove(); eng.sync()
keep = {k: np.zeros_like(h(getattr(wl, k))) for k in ("xcoms", "ycoms", "pi", "theta")}
okbuf = np.zeros(N, dtype=np.uint8)
def both(tag):
    for rep in range(3):
        if rep == 2:
            os.environ["GS_PIPE_TRACE"] = "1"; sys.stderr.write("==== %s prove\n" % tag); sys.stderr.flush()
        t0 = time.perf_counter()
        eng.prove_batch(0, N, 4, 4, X, Y, A, B, G, R, S, T, out=keep)
        t1 = time.perf_counter()
        if rep == 2:
            sys.stderr.write("==== %s verify\n" % tag); sys.stderr.flush()
        eng.verify_batch(0, N, 4, 4, A, B, G, tgt, keep["xcoms"], keep["ycoms"], keep["pi"], keep["theta"], ok=okbuf)
        t2 = time.perf_counter()
        os.environ.pop("GS_PIPE_TRACE", None)
    sys.stderr.write("%s: prove %.1f verify %.1f ms\n" % (tag, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
both("staged")
regs = [X, Y, A, B, G, R, S, T, tgt, okbuf] + list(keep.values())
for a in regs: eng.host_register(a)
both("pinned")

# where the host path's extra GPU time sits: per-kernel times (HIP events around each launch) of the device-resident
# and of the host-pointer verify / prove
def kernel_times(fn):
    eng.prof_enable(True)
    eng.prof_reset()
    fn()
    eng.sync()
    r = {k: ms for k, ms, _ in eng.prof_get()}
    eng.prof_enable(False)
    return r

for tag, dev_fn, host_fn in (
        ("prove", lambda: wl.prove(), lambda: eng.prove_batch(0, N, 4, 4, X, Y, A, B, G, R, S, T, out=keep)),
        ("verify", lambda: wl.verify(), lambda: eng.verify_batch(0, N, 4, 4, A, B, G, tgt, keep["xcoms"], keep["ycoms"],
                                                                  keep["pi"], keep["theta"], ok=okbuf))):
    d, hh = kernel_times(dev_fn), kernel_times(host_fn)
    sys.stderr.write("%s kernels (device-resident / host pinned, ms):\n" % tag)
    for k in hh:
        sys.stderr.write("   %-28s %8.2f %8.2f\n" % (k, d.get(k, float("nan")), hh[k]))
    sys.stderr.write("   %-28s %8.2f %8.2f\n" % ("sum", sum(d.values()), sum(hh.values())))
