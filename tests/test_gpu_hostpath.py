"""The host-pointer entry points -- what a drop-in caller of prove.rs:29-52 / verifier.rs:18-21 holds -- go through the
pinned staging pipeline (csrc/gs_amd.hip: HostPipe): memcpy workers, a copy stream, per-array events, kernels gated on
the arrays they read.  Overlap must never change a byte: at 2^14 equations (several 2 MB pieces per array, uploads
still in flight when the first kernels start) the host path's commitments and proofs equal the device-resident path's
and the C oracle's on sampled equations, the verdicts are exact, and a second call on the same context (staging
buffers reused) gives the same bytes.  Small ragged batches of every type on both curves cover arrays shorter than one
piece and the scalar-valued variants whose variables are read by the preparation kernel."""
import os
import sys

import numpy as np
import pytest

from gsutil import REPO

sys.path.insert(0, os.path.join(REPO, "oracle"))

pytestmark = pytest.mark.gpu


def _host_arrays(wl):
    host = lambda t: t.cpu().numpy()
    return [host(getattr(wl, k)) for k in ("X", "Y", "A", "B", "Gamma", "R", "S", "T", "target")]


def test_host_path_2p14_matches_device_path_and_oracle():
    import groth_sahai_rs_amd as gs
    import gs_ref_py as ref
    from gpubatch import oracle_check
    from groth_sahai_rs_amd.workload import Workload

    N, m, n = 1 << 14, 4, 4
    eng = gs.Engine(0, 0)
    wl = Workload(eng, ty=0, N=N, m=m, n=n, seed=4141, corrupt_every=0)
    wl.prove()
    eng.sync()
    X, Y, A, B, G, R, S, T, tgt = _host_arrays(wl)
    want = {k: getattr(wl, k).cpu().numpy() for k in ("xcoms", "ycoms", "pi", "theta")}
    for rep in range(2):  # the second call reuses the pinned and device staging of the first
        got = eng.prove_batch(0, N, m, n, X, Y, A, B, G, R, S, T)
        for k in want:
            assert (got[k] == want[k]).all(), (k, rep)
    sample = [0, 1, 63, 64, 4095, 4096, 8191, 8192, 12345, N - 1] + list(range(7000, 7006))
    oracle_check(ref, "bls12_381", eng, wl, sample,
                 host_arrays=(X, Y, A, B, G, R, S, T, got["xcoms"], got["ycoms"], got["pi"], got["theta"], tgt))
    ok = eng.verify_batch(0, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], got["pi"], got["theta"])
    assert ok.all()
    bad_pi, bad_th = got["pi"].copy(), got["theta"].copy()
    per_pi, per_th = len(bad_pi) // N, len(bad_th) // N
    for e in (0, 77, N - 1):
        bad_pi[e * per_pi + 9] ^= 1
    bad_th[5000 * per_th + 3] ^= 8
    ok = eng.verify_batch(0, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], bad_pi, bad_th)
    wantok = np.ones(N, dtype=np.uint8)
    wantok[[0, 77, 5000, N - 1]] = 0
    assert (ok == wantok).all()
    # prove only (no commitments wanted): NULL output arrays are skipped by the pipeline
    got2 = eng.prove_batch(0, N, m, n, X, Y, A, B, G, R, S, T, want_coms=False)
    assert got2["xcoms"] is None and (got2["pi"] == want["pi"]).all() and (got2["theta"] == want["theta"]).all()
    eng.close()


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
@pytest.mark.parametrize("ty", [0, 1, 2, 3])
def test_host_path_small_ragged_all_types(cname, cid, ty):
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    N, m, n = 67, 3, 5
    eng = gs.Engine(cid, 0)
    wl = Workload(eng, ty=ty, N=N, m=m, n=n, seed=4200 + ty, corrupt_every=0)
    wl.prove()
    eng.sync()
    X, Y, A, B, G, R, S, T, tgt = _host_arrays(wl)
    want = {k: getattr(wl, k).cpu().numpy() for k in ("xcoms", "ycoms", "pi", "theta")}
    got = eng.prove_batch(ty, N, m, n, X, Y, A, B, G, R, S, T)
    for k in want:
        assert (got[k] == want[k]).all(), k
    ok = eng.verify_batch(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], got["pi"], got["theta"])
    assert ok.all()
    bad = got["theta"].copy()
    bad[66 * (len(bad) // N) + 1] ^= 2
    ok = eng.verify_batch(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], got["pi"], bad)
    assert ok[:66].all() and ok[66] == 0
    eng.close()


def test_host_path_page_locked_arrays():
    """Arrays the caller has page-locked with gs_host_register are moved by DMA directly, the others go through the
    staging copy -- in one call, array by array (one of them lives in memory another allocator has page-locked, a torch
    pinned tensor: that is NOT taken as registered, see include/gs_amd.h).  Bytes and verdicts equal the device-resident
    path's whichever way each array travels, and after gs_host_unregister the same buffers are staged again."""
    import torch

    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    N, m, n = 1 << 13, 4, 4
    eng = gs.Engine(0, 0)
    wl = Workload(eng, ty=0, N=N, m=m, n=n, seed=4343, corrupt_every=0)
    wl.prove()
    eng.sync()
    def fresh(a):  # buffers that get registered sit on mappings of their own (capi.Engine.host_buffer)
        b = eng.host_buffer(a.nbytes)
        b[:] = a.reshape(-1).view(np.uint8)
        return b

    X, Y, A, B, G, R, S, T, tgt = [fresh(a) for a in _host_arrays(wl)]
    want = {k: getattr(wl, k).cpu().numpy() for k in ("xcoms", "ycoms", "pi", "theta")}
    out = {k: eng.host_buffer(v.nbytes) for k, v in want.items()}
    okbuf = eng.host_buffer(N)
    # A lives in hipHostMalloc memory (torch's pinned allocator: staged), the others are registered in two steps
    A_pin = torch.from_numpy(np.array(A)).pin_memory()
    A = A_pin.numpy()
    first = [X, G, out["pi"], out["xcoms"]]
    rest = [Y, B, R, S, T, tgt, out["ycoms"], out["theta"], okbuf]
    wantok = np.ones(N, dtype=np.uint8)
    wantok[[3, N - 1]] = 0

    def check(tag):
        for v in out.values():
            v[:] = 0
        eng.prove_batch(0, N, m, n, X, Y, A, B, G, R, S, T, out=out)
        for k in want:
            assert (out[k] == want[k]).all(), (tag, k)
        ok = eng.verify_batch(0, N, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], out["theta"], ok=okbuf)
        assert ok is okbuf and ok.all(), tag
        per = out["pi"].size // N
        out["pi"][3 * per + 5] ^= 1
        out["pi"][(N - 1) * per + 100] ^= 4
        ok = eng.verify_batch(0, N, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], out["theta"], ok=okbuf)
        assert (ok == wantok).all(), tag

    check("one pinned array")
    for a in first:
        eng.host_register(a)
    check("some registered")
    for a in rest:
        eng.host_register(a)
    check("all registered")
    with pytest.raises(gs.GsError):  # registered twice
        eng.host_register(X)
    for a in first + rest:
        eng.host_unregister(a)
    with pytest.raises(gs.GsError):  # not registered any more
        eng.host_unregister(X)
    check("unregistered again")
    eng.close()


def test_host_alloc_buffers_take_the_dma_direct_path():
    """gs_host_alloc: pre-faulted, page-locked, registered buffers in one call.  Every array of a
    prove + verify lives in such memory: same bytes and verdicts as the device-resident path, the buffers are known to
    the registry (a second registration of one is refused), gs_host_free releases them (twice is an error) and a
    freed range is staged again if the caller still uses memory at that address."""
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    N, m, n = 1 << 12, 4, 4
    eng = gs.Engine(0, 0)
    wl = Workload(eng, ty=0, N=N, m=m, n=n, seed=4345, corrupt_every=0)
    wl.prove()
    eng.sync()

    def alloc(a):
        b = eng.host_alloc(a.nbytes)
        b[:] = a.reshape(-1).view(np.uint8)
        return b

    X, Y, A, B, G, R, S, T, tgt = [alloc(a) for a in _host_arrays(wl)]
    want = {k: getattr(wl, k).cpu().numpy() for k in ("xcoms", "ycoms", "pi", "theta")}
    out = {k: eng.host_alloc(v.nbytes) for k, v in want.items()}
    okbuf = eng.host_alloc(N)
    eng.prove_batch(0, N, m, n, X, Y, A, B, G, R, S, T, out=out)
    for k in want:
        assert (out[k] == want[k]).all(), k
    per = out["pi"].size // N
    out["pi"][5 * per + 3] ^= 2
    ok = eng.verify_batch(0, N, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], out["theta"], ok=okbuf)
    assert ok is okbuf and ok[5] == 0 and ok.sum() == N - 1
    with pytest.raises(gs.GsError):
        eng.host_register(X)  # already registered (by gs_host_alloc)
    import ctypes

    x_addr = X.ctypes.data
    del X  # (the view dies with the mapping)
    eng._chk(eng.lib.gs_host_free(eng.ctx, ctypes.c_void_p(x_addr)))
    eng._allocs.pop(x_addr, None)
    with pytest.raises(gs.GsError):
        eng._chk(eng.lib.gs_host_free(eng.ctx, ctypes.c_void_p(x_addr)))  # freed already
    eng.close()  # frees the rest
