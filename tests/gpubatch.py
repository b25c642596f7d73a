"""Shared driver of the batch-scale GPU parity tests: build a synthetic batch on the GPU (Workload), prove it through
the C ABI, compare sampled equations bit for bit with the C restatement of the reference path (oracle/gs_ref.c:
commitments, pi, theta and the verdict), then put the WHOLE batch through the size-independent properties (every honest
proof verifies, every corrupted one is rejected, batched and exact verdicts agree).

`opts` are planner overrides (gs_set_option) that force a particular kernel shape; `expect` lists kernel names that
must have run (checked through the library's HIP-event profile of a second, identical pass, whose outputs must equal
the first's)."""
import fnmatch
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from gsutil import REPO

sys.path.insert(0, os.path.join(REPO, "oracle"))

_POOL = None


def pool():
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=max(2, min(16, (os.cpu_count() or 4))))
    return _POOL


def oracle_check(ref, cname, eng, wl, sample, host_arrays=None):
    """Every equation index in `sample`: oracle outputs == GPU outputs, oracle verdict on them == 1."""
    host = lambda t: t.cpu().numpy()
    ty, m, n, sh = wl.ty, wl.m, wl.n, wl.sh
    kx, ky, sx, sy, st = sh["kx"], sh["ky"], sh["sx"], sh["sy"], sh["st"]
    if host_arrays is None:
        host_arrays = tuple(map(host, (wl.X, wl.Y, wl.A, wl.B, wl.Gamma, wl.R, wl.S, wl.T, wl.xcoms, wl.ycoms, wl.pi,
                                       wl.theta, wl.target)))
    X, Y, A, B, G, R, S, T, xc, yc, pi, th, tgt = host_arrays
    cut = lambda a, e, sz: a[e * sz:(e + 1) * sz]

    def one(e):
        out = ref.commit_and_prove(cname, ty, m, n, cut(X, e, m * sx), cut(Y, e, n * sy), cut(A, e, n * sx),
                                   cut(B, e, m * sy), cut(G, e, m * n * 32), cut(R, e, m * kx * 32),
                                   cut(S, e, n * ky * 32), cut(T, e, ky * kx * 32), wl.crs)
        for name, got, per in (("xcoms", xc, m * eng.COM1), ("ycoms", yc, n * eng.COM2), ("pi", pi, kx * eng.COM2),
                               ("theta", th, ky * eng.COM1)):
            if not (out[name] == cut(got, e, per)).all():
                return (e, name)
        v = ref.verify(cname, ty, m, n, cut(A, e, n * sx), cut(B, e, m * sy), cut(G, e, m * n * 32), cut(tgt, e, st),
                       out["xcoms"], out["ycoms"], out["pi"], out["theta"], wl.crs)
        return None if v == 1 else (e, "oracle verdict")

    bad = [r for r in pool().map(one, list(sample)) if r is not None]
    assert not bad, (cname, ty, bad[:4])
    return host_arrays


def run_batch(curve_id, cname, ty, N, m, n, sample, opts=None, expect=None, seed=None, corrupt_every=None, rlc=True):
    import groth_sahai_rs_amd as gs
    import gs_ref_py as ref
    from groth_sahai_rs_amd.workload import Workload

    eng = gs.Engine(curve_id, 0)
    for k, v in (opts or {}).items():
        eng.set_option(k, v)
    wl = Workload(eng, ty=ty, N=N, m=m, n=n, seed=777 + ty if seed is None else seed,
                  corrupt_every=max(N // 4, 1) if corrupt_every is None else corrupt_every)
    wl.prove()
    eng.sync()
    arrays = oracle_check(ref, cname, eng, wl, sample)
    xc, yc, pi, th = arrays[8:12]
    # whole batch: exact verdicts
    wl.verify()
    eng.sync()
    assert wl.ok.cpu().numpy().all()
    if rlc:
        assert eng.gt_finalize(wl.verify_rlc().cpu().numpy()) == 1
    if expect:
        # the same pass under the library's per-kernel profile: names of what ran, and identical outputs
        eng.prof_enable(True)
        eng.prof_reset()
        wl.prove()
        wl.verify()
        eng.sync()
        names = [p[0] for p in eng.prof_get()]
        eng.prof_enable(False)
        for want in expect:  # exact kernel name, or a pattern where the planner may pick a sub-variant
            assert any(fnmatch.fnmatchcase(nm, want) for nm in names), (want, names)
        for a, t in ((xc, wl.xcoms), (yc, wl.ycoms), (pi, wl.pi), (th, wl.theta)):
            assert (t.cpu().numpy() == a).all()
        assert wl.ok.cpu().numpy().all()
    bad = wl.corrupt()
    wl.verify()
    eng.sync()
    ok = wl.ok.cpu().numpy()
    badset = set(bad)
    want = np.ones(N, dtype=np.uint8)
    want[list(badset)] = 0
    assert (ok == want).all(), ("verdicts", np.nonzero(ok != want)[0][:8])
    if rlc and bad:
        assert eng.gt_finalize(wl.verify_rlc().cpu().numpy()) == 0
    if bad:  # the C oracle agrees on one corrupted proof
        sh = wl.sh
        kx, ky, sx, sy, st = sh["kx"], sh["ky"], sh["sx"], sh["sy"], sh["st"]
        cut = lambda a, e, sz: a[e * sz:(e + 1) * sz]
        A, B, G, tgt = arrays[2], arrays[3], arrays[4], arrays[12]
        e = bad[0]
        assert ref.verify(cname, ty, m, n, cut(A, e, n * sx), cut(B, e, m * sy), cut(G, e, m * n * 32), cut(tgt, e, st),
                          cut(xc, e, m * eng.COM1), cut(yc, e, n * eng.COM2),
                          cut(wl.pi.cpu().numpy(), e, kx * eng.COM2), cut(th, e, ky * eng.COM1), wl.crs) == 0
    eng.close()
