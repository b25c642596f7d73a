"""The multi-GPU layer of the C ABI (gs_ctx_create_multi, include/gs_amd.h) on the one GPU a test box has: ndev = 1
through the multi entry points must give the single-device bytes, the block partition must tile the batch, and the
batched verifier's accumulator exchange must really go through RCCL (one rank here; the 8-GPU run is the driver's).
Unmeasured on multi-GPU hardware until a SCALE line exists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_multi_entry_matches_single_device_bytes():
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    N, m, n = 37, 3, 2
    for ty in (0, 1, 2, 3):
        eng = gs.Engine(0, 0)
        wl = Workload(eng, ty=ty, N=N, m=m, n=n, seed=8800 + ty, corrupt_every=0)
        wl.prove()
        eng.sync()
        host = lambda t: t.cpu().numpy()
        X, Y, A, B, G, R, S, T, tgt = map(host, (wl.X, wl.Y, wl.A, wl.B, wl.Gamma, wl.R, wl.S, wl.T, wl.target))
        want = {k: host(getattr(wl, k)) for k in ("xcoms", "ycoms", "pi", "theta")}
        me = gs.MultiEngine(0, [0])
        me.set_crs(wl.crs)
        # the partition tiles [0, N)
        assert me.shard(N, 0) == (0, N)
        got = me.prove_batch(ty, N, m, n, X, Y, A, B, G, R, S, T)
        for k in want:
            assert (got[k] == want[k]).all(), (ty, k)
        ok = me.verify_batch(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], got["pi"], got["theta"])
        assert ok.all()
        rho = np.frombuffer(np.random.default_rng(ty).bytes(N * 32), dtype=np.uint64) | np.uint64(1)
        v, pairs = me.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], got["pi"], got["theta"], rho)
        assert v == 1 and me.uses_rccl(), "the accumulator pairs must travel through RCCL"
        # same accumulator pair as the single-device batched verifier with the same rho
        v1, acc1 = eng.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], got["pi"], got["theta"], rho)
        assert v1 == 1 and (acc1 == pairs).all()
        bad = got["pi"].copy()
        bad[5] ^= 4
        ok = me.verify_batch(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], bad, got["theta"])
        assert ok[0] == 0 and ok[1:].all()
        v, _ = me.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], bad, got["theta"], rho)
        assert v == 0
        me.close()
        eng.close()


def test_multi_rejects_bad_device_lists_and_shapes():
    import ctypes

    import groth_sahai_rs_amd as gs

    lib = gs.load_library()
    h = ctypes.c_void_p()
    two = (ctypes.c_int * 2)(0, 0)
    assert lib.gs_ctx_create_multi(0, two, 2, ctypes.byref(h)) == 3       # the same device twice
    assert lib.gs_ctx_create_multi(0, two, 0, ctypes.byref(h)) == 3       # no device
    far = (ctypes.c_int * 1)(99)
    assert lib.gs_ctx_create_multi(0, far, 1, ctypes.byref(h)) == 2       # GS_ERR_DEVICE
    me = gs.MultiEngine(0, [0])
    with pytest.raises(gs.GsError) as ei:
        z = np.zeros(8, dtype=np.uint8)
        me.verify_batch(0, 1, 1, 1, z, z, z, z, z, z, z, z)
    assert ei.value.code == 1
    me.close()
