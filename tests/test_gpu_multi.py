"""The multi-GPU layer of the C ABI (gs_ctx_create_multi, include/gs_amd.h) on the one GPU a test box has.
  * ndev = 1 through the multi entry points gives the single-device bytes;
  * GS_MULTI_SHARED_DEVICES puts 2, 3 and 8 SHARDS on that one GPU, so the block partition with lo > 0, every
    type-dependent offset of the host entry points, empty blocks (N < ndev), the per-shard device-pointer family and the
    accumulator-pair exchange + shard-order product of the batched verifier all execute here, before an 8-GPU node
    runs them: same bytes and verdicts as the single-device engine, a corrupted proof in the LAST shard found by both
    verifiers, rho indexed globally.
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
        assert v == 1 and not me.uses_rccl() and "one shard" in me.exchange_note()  # nothing to exchange with one shard
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
    assert lib.gs_ctx_create_multi(0, two, 2, ctypes.byref(h)) == 3       # the same device twice (without the flag)
    assert lib.gs_ctx_create_multi(0, two, 0, ctypes.byref(h)) == 3       # no device
    far = (ctypes.c_int * 1)(99)
    assert lib.gs_ctx_create_multi(0, far, 1, ctypes.byref(h)) == 2       # GS_ERR_DEVICE
    me = gs.MultiEngine(0, [0])
    with pytest.raises(gs.GsError) as ei:
        z = np.zeros(8, dtype=np.uint8)
        me.verify_batch(0, 1, 1, 1, z, z, z, z, z, z, z, z)
    assert ei.value.code == 1
    me.close()


@pytest.mark.parametrize("ndev,N", [(2, 37), (3, 64), (8, 5)])
@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
def test_split_over_several_shards_of_one_gpu(cname, cid, ndev, N):
    import torch

    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    m, n = 3, 2
    host = lambda t: t.cpu().numpy()
    for ty in (0, 1, 2, 3):
        eng = gs.Engine(cid, 0)
        wl = Workload(eng, ty=ty, N=N, m=m, n=n, seed=8900 + 10 * ndev + ty, corrupt_every=0)
        wl.prove()
        eng.sync()
        names = ("X", "Y", "A", "B", "Gamma", "R", "S", "T", "target")
        dev = {k: getattr(wl, k) for k in names}
        X, Y, A, B, G, R, S, T, tgt = [host(dev[k]) for k in names]
        want = {k: host(getattr(wl, k)) for k in ("xcoms", "ycoms", "pi", "theta")}
        me = gs.MultiEngine(cid, [0] * ndev, shared_devices=True)
        me.set_crs(wl.crs)
        blocks = [me.shard(N, i) for i in range(ndev)]
        assert blocks[0][0] == 0 and blocks[-1][1] == N and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
        assert (N >= ndev) or any(h == l for l, h in blocks)  # 5 equations on 8 shards: empty blocks
        # ---- host-pointer family: whole batch in, whole batch out
        got = me.prove_batch(ty, N, m, n, X, Y, A, B, G, R, S, T)
        for k in want:
            assert (got[k] == want[k]).all(), (ty, k)
        args = (A, B, G, tgt, got["xcoms"], got["ycoms"])
        assert me.verify_batch(ty, N, m, n, *args, got["pi"], got["theta"]).all()
        bad = got["pi"].copy()
        bad[(N - 1) * (len(bad) // N) + 5] ^= 4  # the LAST equation: it lives in the last non-empty shard
        ok_multi = me.verify_batch(ty, N, m, n, *args, bad, got["theta"])
        ok_single = eng.verify_batch(ty, N, m, n, *args, bad, got["theta"])
        assert (ok_multi == ok_single).all() and ok_multi[:-1].all() and ok_multi[-1] == 0
        rho = np.frombuffer(np.random.default_rng(100 + ty).bytes(N * 32), dtype=np.uint64) | np.uint64(1)
        v, pairs = me.verify_batch_rlc(ty, N, m, n, *args, got["pi"], got["theta"], rho)
        assert v == 1 and "share a device" in me.exchange_note()
        pairs = pairs.reshape(ndev, -1)
        one = None
        for i, (lo, hi) in enumerate(blocks):  # every shard's pair = the single-device pair of ITS block with ITS rho
            if hi == lo:
                one = pairs[i] if one is None else one
                assert (pairs[i] == one).all() and (pairs[i][:eng.GT] == pairs[i][eng.GT:]).all()  # (1, 1)
                continue
            sl = lambda a, per: a[lo * per:hi * per]
            per = lambda a: len(a) // N
            v1, acc1 = eng.verify_batch_rlc(ty, hi - lo, m, n, sl(A, per(A)), sl(B, per(B)), sl(G, per(G)),
                                            sl(tgt, per(tgt)), sl(got["xcoms"], per(got["xcoms"])),
                                            sl(got["ycoms"], per(got["ycoms"])), sl(got["pi"], per(got["pi"])),
                                            sl(got["theta"], per(got["theta"])), rho[4 * lo:4 * hi])
            assert v1 == 1 and (acc1 == pairs[i]).all(), (ty, i)
        assert eng.gt_finalize(pairs.reshape(-1)) == 1
        v, _ = me.verify_batch_rlc(ty, N, m, n, *args, bad, got["theta"], rho)
        assert v == 0
        # ---- device-pointer family: shard i's block already on its device
        cut = lambda t, i: t[blocks[i][0] * (t.numel() // N):blocks[i][1] * (t.numel() // N)].clone()
        per_shard = lambda t: [cut(t, i) for i in range(ndev)]
        outs = {k: [torch.zeros_like(cut(getattr(wl, k), i)) for i in range(ndev)] for k in ("xcoms", "ycoms", "pi", "theta")}
        ins = {k: per_shard(dev[k]) for k in names}
        me.prove_batch_dev(ty, N, m, n, ins["X"], ins["Y"], ins["A"], ins["B"], ins["Gamma"], ins["R"], ins["S"], ins["T"],
                           outs["xcoms"], outs["ycoms"], outs["pi"], outs["theta"])
        me.sync()
        for k in want:
            assert (np.concatenate([host(t) for t in outs[k]]) == want[k]).all(), (ty, k, "dev")
        okd = [torch.zeros(hi - lo, dtype=torch.uint8, device="cuda:0") for lo, hi in blocks]
        me.verify_batch_dev(ty, N, m, n, ins["A"], ins["B"], ins["Gamma"], ins["target"], outs["xcoms"], outs["ycoms"],
                            outs["pi"], outs["theta"], okd)
        me.sync()
        assert np.concatenate([host(t) for t in okd]).all()
        rho_t = torch.from_numpy(rho.view(np.int64)).to("cuda:0")
        rho_sh = [rho_t[4 * lo:4 * hi].clone() for lo, hi in blocks]
        vd, pairs_d = me.verify_batch_rlc_dev(ty, N, m, n, ins["A"], ins["B"], ins["Gamma"], ins["target"], outs["xcoms"],
                                              outs["ycoms"], outs["pi"], outs["theta"], rho_sh)
        assert vd == 1 and (pairs_d.reshape(ndev, -1) == pairs).all()  # the same pairs as the host family
        last = [i for i, (lo, hi) in enumerate(blocks) if hi > lo][-1]
        outs["pi"][last][-7] ^= 2  # a proof of the last non-empty shard
        vd, _ = me.verify_batch_rlc_dev(ty, N, m, n, ins["A"], ins["B"], ins["Gamma"], ins["target"], outs["xcoms"],
                                        outs["ycoms"], outs["pi"], outs["theta"], rho_sh)
        assert vd == 0
        me.close()
        eng.close()


def test_pair_exchange_fallback_chain():
    """The accumulator-pair exchange of the batched verifier walks three legs (RCCL all-gather, device / peer copies,
    host-staged copies); a leg that fails AT RUN TIME is marked failed and the same call retries on the next one
    (the first 8-GPU run must not die on a failed collective).  Each failure is injected
    ("exchange_fail" bit k: leg k fails when it runs) on three shards of the one GPU: same verdicts and the same
    gathered pairs on every leg, the note names the leg used and why the earlier ones failed, a failed leg is not
    tried again, a corrupted proof is still found, and only when EVERY leg fails does the call return an error."""
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    N, m, n, ty, nd = 19, 2, 3, 0, 3
    eng = gs.Engine(0, 0)
    wl = Workload(eng, ty=ty, N=N, m=m, n=n, seed=8950, corrupt_every=0)
    wl.prove()
    eng.sync()
    host = lambda t: t.cpu().numpy()
    A, B, G, tgt, xc, yc, pi, th = [host(getattr(wl, k)) for k in ("A", "B", "Gamma", "target", "xcoms", "ycoms", "pi", "theta")]
    rho = np.frombuffer(np.random.default_rng(77).bytes(N * 32), dtype=np.uint64) | np.uint64(1)
    bad = pi.copy()
    bad[(N - 1) * (len(bad) // N) + 9] ^= 8
    me = gs.MultiEngine(0, [0] * nd, shared_devices=True)
    me.set_crs(wl.crs)
    v, want = me.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, xc, yc, pi, th, rho)
    assert v == 1 and "device / peer copies" in me.exchange_note() and "share a device" in me.exchange_note()
    for mask, leg, failed in ((1, "device / peer copies", ["rccl all-gather"]),
                              (3, "host-staged copies", ["rccl all-gather", "device / peer copies"]),
                              (2, "host-staged copies", ["device / peer copies"])):
        me.set_option("exchange_reset", 1)
        me.set_option("exchange_fail", mask)
        v, pairs = me.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, xc, yc, pi, th, rho)
        note = me.exchange_note()
        assert v == 1 and (pairs == want).all(), (mask, note)
        assert "last exchange: " + leg in note, note
        for f in failed:
            assert f + " failed earlier: injected failure" in note, note
        # sticky: with the hook off again the failed legs stay out of the chain (no retry of a leg that has failed)
        me.set_option("exchange_fail", 0)
        v, pairs = me.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, xc, yc, bad, th, rho)
        assert v == 0 and "last exchange: " + leg in me.exchange_note()
    me.set_option("exchange_reset", 1)
    me.set_option("exchange_fail", 7)
    with pytest.raises(gs.GsError) as ei:
        me.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, xc, yc, pi, th, rho)
    assert ei.value.code == 2 and "every leg failed" in str(ei.value)
    me.set_option("exchange_reset", 1)
    me.set_option("exchange_fail", 0)
    me.set_option("exchange_leg", 2)  # start at the host-staged leg
    v, pairs = me.verify_batch_rlc(ty, N, m, n, A, B, G, tgt, xc, yc, pi, th, rho)
    assert v == 1 and (pairs == want).all() and "last exchange: host-staged copies" in me.exchange_note()
    me.close()
    eng.close()


def test_multi_rlc_rejects_null_rho():
    import ctypes

    import groth_sahai_rs_amd as gs

    me = gs.MultiEngine(0, [0, 0], shared_devices=True)
    z = np.zeros(16, dtype=np.uint8)
    ok = np.zeros(1, dtype=np.uint8)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    rc = me.lib.gs_multi_verify_batch_rlc(me.h, 0, ctypes.c_size_t(4), 1, 1, p(z), p(z), p(z), p(z), p(z), p(z), p(z), p(z),
                                          ctypes.c_void_p(0), ctypes.c_void_p(0), p(ok))
    assert rc == 3 and "rho" in me.lib.gs_multi_last_error(me.h).decode()
    me.close()


def test_multi_shards_read_sub_ranges_of_registered_arrays():
    """The host entry points of the multi layer hand every shard a SUB-RANGE of the caller's arrays.  With the arrays
    page-locked through gs_host_register (a per-process list: any context may register), every shard's uploads go by
    DMA straight out of its sub-range -- the look-up is by containment, on the shards' own threads at once.  Same bytes
    and verdicts as the single-device engine."""
    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.workload import Workload

    N, m, n, ty = 203, 3, 4, 0
    eng = gs.Engine(0, 0)
    wl = Workload(eng, ty=ty, N=N, m=m, n=n, seed=8891, corrupt_every=0)
    wl.prove()
    eng.sync()
    host = lambda t: t.cpu().numpy()

    def fresh(a):
        b = eng.host_buffer(a.nbytes)
        b[:] = a.reshape(-1).view(np.uint8)
        return b

    X, Y, A, B, G, R, S, T, tgt = [fresh(host(getattr(wl, k))) for k in ("X", "Y", "A", "B", "Gamma", "R", "S", "T", "target")]
    want = {k: host(getattr(wl, k)) for k in ("xcoms", "ycoms", "pi", "theta")}
    regs = [X, Y, A, B, G, R, S, T, tgt]
    for a in regs:
        eng.host_register(a)
    me = gs.MultiEngine(0, [0, 0, 0], shared_devices=True)
    me.set_crs(wl.crs)
    got = me.prove_batch(ty, N, m, n, X, Y, A, B, G, R, S, T)
    for k in want:
        assert (got[k] == want[k]).all(), k
    bad = got["theta"].copy()
    bad[(N - 1) * (bad.size // N) + 7] ^= 16  # last shard
    ok = me.verify_batch(ty, N, m, n, A, B, G, tgt, got["xcoms"], got["ycoms"], got["pi"], bad)
    assert ok[:-1].all() and ok[-1] == 0
    me.close()
    for a in regs:
        eng.host_unregister(a)
    eng.close()
