"""GPU parity tests: the HIP path, called through the C ABI, against the golden
fixtures produced by the big-integer oracle (tests/golden/*.json).  Statements
follow the reference's own tests (tests/prover.rs:25-172)."""
import numpy as np
import pytest

from gsutil import curve

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    import groth_sahai_rs_amd as gs

    out = {}
    for cname, cid in (("bls12_381", 0), ("bn254", 1)):
        c = curve(cname)
        e = gs.Engine(cid, 0)
        g = c.golden["crs"]
        crs = np.concatenate([c.com1(g["u"][0]), c.com1(g["u"][1]), c.com2(g["v"][0]), c.com2(g["v"][1]),
                              c.g1(g["g1"]), c.g2(g["g2"]), c.f12(g["gt"])])
        e.set_crs(crs)
        out[cname] = (c, e)
    return out


CURVES = ["bls12_381", "bn254"]


def enc_side(c, ty, side, vals):
    xg = ty in (0, 1)
    yg = ty in (0, 2)
    if side == "x":
        return np.concatenate([c.g1(v) if xg else c.fr_hex(v) for v in vals])
    return np.concatenate([c.g2(v) if yg else c.fr_hex(v) for v in vals])


def enc_target(c, ty, t):
    return {0: c.f12, 1: c.g1, 2: c.g2, 3: c.fr_hex}[ty](t)


@pytest.mark.parametrize("cname", CURVES)
def test_smul_and_pairing_hooks(engines, cname):
    c, e = engines[cname]
    g = c.golden
    g1 = c.g1(g["g1_smul"][0]["out"])
    g2 = c.g2(g["g2_smul"][0]["out"])
    ks = np.stack([c.fr_hex(x["k"]) for x in g["g1_smul"]] + [c.fr(0)])
    o1 = e.g_mul_batch(1, g1, ks, broadcast=True)
    o2 = e.g_mul_batch(2, g2, ks, broadcast=True)
    for i, x in enumerate(g["g1_smul"]):
        assert c.g1_dec(o1[i].view(np.uint64)) == x["out"]
        assert c.g2_dec(o2[i].view(np.uint64)) == g["g2_smul"][i]["out"]
    assert not o1[-1].any() and not o2[-1].any()
    for pe in g["pairing"]:
        out = e.multi_pairing_batch(1, 1, c.g1(pe["p"]), c.g2(pe["q"]))
        assert c.f12_dec(out[0].view(np.uint64)) == pe["out"]
    ps = g["pairing_sum"]
    x = np.concatenate([c.com1(v) for v in ps["x"]])
    y = np.concatenate([c.com2(v) for v in ps["y"]])
    out = e.pairing_sum(len(ps["x"]), x, y)
    for cell in range(4):
        assert c.f12_dec(out[cell].view(np.uint64)) == ps["out"][cell], cell


@pytest.mark.parametrize("cname", CURVES)
def test_left_mul_hook(engines, cname):
    c, e = engines[cname]
    lm = c.golden["left_mul"]
    lhs = c.fr_mat(lm["lhs"])
    rows, k = len(lm["lhs"]), len(lm["lhs"][0])
    o1 = e.mat_left_mul(1, rows, k, lhs, np.concatenate([c.com1(v) for v in lm["com1"]]))
    o2 = e.mat_left_mul(2, rows, k, lhs, np.concatenate([c.com2(v) for v in lm["com2"]]))
    for i in range(rows):
        got1 = o1[i].view(np.uint64).reshape(2, -1)
        got2 = o2[i].view(np.uint64).reshape(2, -1)
        assert [c.g1_dec(got1[0]), c.g1_dec(got1[1])] == lm["out1"][i]
        assert [c.g2_dec(got2[0]), c.g2_dec(got2[1])] == lm["out2"][i]


@pytest.mark.parametrize("cname", CURVES)
def test_commit_prove_verify_golden(engines, cname):
    """commit_and_prove bit-exact vs the oracle for every golden case, then
    verify == oracle verdict, then the negative twins."""
    c, e = engines[cname]
    for case in c.golden["cases"]:
        ty, m, n = case["type"], case["m"], case["n"]
        X = enc_side(c, ty, "x", case["xvars"])
        Y = enc_side(c, ty, "y", case["yvars"])
        A = enc_side(c, ty, "x", case["a"])
        B = enc_side(c, ty, "y", case["b"])
        G = c.fr_mat(case["gamma"])
        R, S, T = c.fr_mat(case["R"]), c.fr_mat(case["S"]), c.fr_mat(case["T"])
        out = e.prove_batch(ty, 1, m, n, X, Y, A, B, G, R, S, T, want_coms=True)
        xc = out["xcoms"].view(np.uint64).reshape(m, 2, -1)
        yc = out["ycoms"].view(np.uint64).reshape(n, 2, -1)
        pi = out["pi"].view(np.uint64).reshape(-1, 2, 4 * c.nq)
        th = out["theta"].view(np.uint64).reshape(-1, 2, 2 * c.nq)
        name = case["name"]
        assert [[c.g1_dec(v[0]), c.g1_dec(v[1])] for v in xc] == case["xcoms"], name
        assert [[c.g2_dec(v[0]), c.g2_dec(v[1])] for v in yc] == case["ycoms"], name
        assert [[c.g2_dec(v[0]), c.g2_dec(v[1])] for v in pi] == case["pi"], name
        assert [[c.g1_dec(v[0]), c.g1_dec(v[1])] for v in th] == case["theta"], name
        # prove-only entry (xcoms/ycoms = NULL) gives the same proof
        out2 = e.prove_batch(ty, 1, m, n, X, Y, A, B, G, R, S, T, want_coms=False)
        assert (out2["pi"] == out["pi"]).all() and (out2["theta"] == out["theta"]).all()
        if "verify" not in case:
            continue
        tgt = enc_target(c, ty, case["target"])
        ok = e.verify_batch(ty, 1, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], out["theta"])
        assert ok[0] == 1, name
        # negative twins of the fixture: theta[0].1 += g1 ; pi[0].0 += g2
        g1 = c.g1(c.golden["crs"]["g1"])
        g2 = c.g2(c.golden["crs"]["g2"])
        one = c.fr(1).reshape(1, -1)
        # compute P + g via the smul hook on 2 points would need an add hook; corrupt limbs instead
        bad = out["theta"].copy()
        bad[c.nq * 8 * 2] ^= 1  # flip a bit of theta[0].1.x
        ok = e.verify_batch(ty, 1, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], bad)
        assert ok[0] == 0, name
        bad = out["pi"].copy()
        bad[3] ^= 0x10
        ok = e.verify_batch(ty, 1, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], bad, out["theta"])
        assert ok[0] == 0, name
        bad = tgt.copy().view(np.uint8)
        bad[0] ^= 1
        ok = e.verify_batch(ty, 1, m, n, A, B, G, bad, out["xcoms"], out["ycoms"], out["pi"], out["theta"])
        assert ok[0] == 0, name


@pytest.mark.parametrize("cname", CURVES)
def test_rlc_batched_verifier(engines, cname):
    """Batched (random-linear-combination) verifier: one final exponentiation per
    batch.  Accepts iff every equation verifies; accumulators of sub-batches
    multiply (what the multi-GPU path all-gathers)."""
    c, e = engines[cname]
    rng = np.random.default_rng(7)
    for ty_cases in ([0, 4], [1, 5], [2, 6], [3, 7]):  # (ref 2x1, dense 2x2) of each type have different shapes
        for idx in ty_cases:
            case = c.golden["cases"][idx]
            ty, m, n = case["type"], case["m"], case["n"]
            X, Y = enc_side(c, ty, "x", case["xvars"]), enc_side(c, ty, "y", case["yvars"])
            A, B = enc_side(c, ty, "x", case["a"]), enc_side(c, ty, "y", case["b"])
            G, R, S, T = (c.fr_mat(case[k]) for k in ("gamma", "R", "S", "T"))
            tgt = enc_target(c, ty, case["target"])
            out = e.prove_batch(ty, 1, m, n, X, Y, A, B, G, R, S, T)
            # batch of 3 copies of the same equation
            rep = lambda a: np.concatenate([np.asarray(a).view(np.uint8).reshape(-1)] * 3)
            args = [rep(A), rep(B), rep(G), rep(tgt), rep(out["xcoms"]), rep(out["ycoms"]), rep(out["pi"]),
                    rep(out["theta"])]
            rho = rng.integers(1, 1 << 63, size=12, dtype=np.uint64)
            ok, acc = e.verify_batch_rlc(ty, 3, m, n, *args, rho)
            assert ok == 1, case["name"]
            # corrupt the middle proof: theta of equation 1
            bad = args[7].copy()
            bad[len(bad) // 3 + 7] ^= 2
            ok2, _ = e.verify_batch_rlc(ty, 3, m, n, *args[:7], bad, rho)
            assert ok2 == 0, case["name"]
            # split: equations {0,1} and {2} on two "ranks": product of accumulators = same verdict
            first = [a[: 2 * (a.size // 3)] for a in args]
            last = [a[2 * (a.size // 3):] for a in args]
            _, acc_a = e.verify_batch_rlc(ty, 2, m, n, *first, rho[:8])
            _, acc_b = e.verify_batch_rlc(ty, 1, m, n, *last, rho[8:])
            assert e.gt_finalize(np.concatenate([acc_a, acc_b])) == 1
            bad_last = [a.copy() for a in last]
            bad_last[6][3] ^= 1
            _, acc_c = e.verify_batch_rlc(ty, 1, m, n, *bad_last, rho[8:])
            assert e.gt_finalize(np.concatenate([acc_a, acc_c])) == 0


@pytest.mark.parametrize("cname", CURVES)
def test_reference_algebraic_properties(engines, cname):
    """Properties the reference's own unit/integration tests assert, through the L2 hooks:
      * "1*4 + 2*5 + 3*6 = 32 in the exponent" for Com matrices      data_structures.rs:1951-2006
      * Com::scalar_mul == component-wise group scalar mul            data_structures.rs:1235-1265
      * iota-map / pairing commutativity for PPE and the scalar maps  tests/commit.rs:22-85
      * pairing with an identity argument is the GT identity          data_structures.rs:1313-1343"""
    c, e = engines[cname]
    crs = c.golden["crs"]
    g1, g2 = c.g1(crs["g1"]), c.g2(crs["g2"])
    frs = lambda vals: np.stack([c.fr(v) for v in vals])
    # k*g for k = 4, 5, 6, 32
    m1 = e.g_mul_batch(1, g1, frs([4, 5, 6, 32, 7]), broadcast=True)
    m2 = e.g_mul_batch(2, g2, frs([4, 5, 6, 32, 7]), broadcast=True)
    col1 = np.concatenate([np.concatenate([m1[i], m1[i]]) for i in range(3)])  # Com1 (kg, kg)
    col2 = np.concatenate([np.concatenate([m2[i], m2[i]]) for i in range(3)])
    lhs = frs([1, 2, 3])
    o1 = e.mat_left_mul(1, 1, 3, lhs, col1)[0]
    o2 = e.mat_left_mul(2, 1, 3, lhs, col2)[0]
    assert (o1 == np.concatenate([m1[3], m1[3]])).all() and (o2 == np.concatenate([m2[3], m2[3]])).all()
    # scalar_mul of a Com element == component-wise smul
    com = np.concatenate([m1[0], m1[1]])  # (4g, 5g)
    out = e.mat_left_mul(1, 1, 1, frs([7]), com)[0].reshape(2, -1)
    want = e.g_mul_batch(1, np.concatenate([m1[0], m1[1]]), frs([7, 7]))
    assert (out[0] == want[0]).all() and (out[1] == want[1]).all()
    # PPE commutativity: pairing(iota1(a1), iota2(a2)) == iota_T(e(a1, a2)) = (1, 1, 1, e(a1,a2))
    a1, a2 = m1[0], m2[1]
    zero1, zero2 = np.zeros_like(a1), np.zeros_like(a2)
    cells = e.pairing_sum(1, np.concatenate([zero1, a1]), np.concatenate([zero2, a2]))
    one = c.f12(["1"] + ["0"] * 11).view(np.uint8)
    at = e.multi_pairing_batch(1, 1, a1, a2)[0]
    assert (cells[0] == one).all() and (cells[1] == one).all() and (cells[2] == one).all() and (cells[3] == at).all()
    # scalar maps: iota2'(y) = y*W2 is the scalar commitment with zero randomness (commit.rs:225-256)
    y, x = 5, 4
    w2y = e.commit("fr_b2", frs([y]), frs([0]))[0]            # y * W2
    w1x = e.commit("fr_b1", frs([x]), frs([0]))[0]            # x * W1
    w2_1 = e.commit("fr_b2", frs([1]), frs([0]))[0]           # W2
    w1_1 = e.commit("fr_b1", frs([1]), frs([0]))[0]           # W1
    # MSMEG1: pairing(iota1(a1), iota2'(y)) == pairing(iota1(y*a1), W2)          (tests/commit.rs:37-52)
    ya1 = e.g_mul_batch(1, a1, frs([y]))[0]
    l = e.pairing_sum(1, np.concatenate([zero1, a1]), w2y)
    r = e.pairing_sum(1, np.concatenate([zero1, ya1]), w2_1)
    assert (l == r).all()
    # MSMEG2: pairing(iota1'(x), iota2(a2)) == pairing(W1, iota2(x*a2))          (tests/commit.rs:54-69)
    xa2 = e.g_mul_batch(2, a2, frs([x]))[0]
    l = e.pairing_sum(1, w1x, np.concatenate([zero2, a2]))
    r = e.pairing_sum(1, w1_1, np.concatenate([zero2, xa2]))
    assert (l == r).all()
    # Quad: pairing(iota1'(x), iota2'(y)) == pairing(W1, (x*y) * W2)              (tests/commit.rs:71-85)
    w2xy = e.commit("fr_b2", frs([x * y]), frs([0]))[0]
    l = e.pairing_sum(1, w1x, w2y)
    r = e.pairing_sum(1, w1_1, w2xy)
    assert (l == r).all()
    # identity arguments give the GT identity; empty sum too
    cells = e.pairing_sum(1, np.concatenate([zero1, zero1]), np.concatenate([a2, a2]))
    assert all((cells[i] == one).all() for i in range(4))
    cells = e.pairing_sum(0, np.zeros(0, np.uint8), np.zeros(0, np.uint8))
    assert all((cells[i] == one).all() for i in range(4))


def test_status_codes_and_empty_batches(engines):
    """The boundary's error behaviour (include/gs_amd.h): status codes where the reference panics (shape asserts,
    prove.rs:106-113) or cannot happen in Rust (null pointers, missing CRS); an empty batch is a no-op."""
    import ctypes

    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.capi import GsError

    c, e = engines["bls12_381"]
    lib = e.lib
    z = ctypes.c_void_p(0)
    one = np.zeros(1 << 16, dtype=np.uint8)
    p = ctypes.c_void_p(one.ctypes.data)
    n0 = ctypes.c_size_t(0)
    n1 = ctypes.c_size_t(1)
    # empty batch: OK without touching any pointer
    assert lib.gs_prove_batch(e.ctx, 0, n0, 4, 4, z, z, z, z, z, z, z, z, z, z, z, z) == 0
    assert lib.gs_verify_batch(e.ctx, 0, n0, 4, 4, z, z, z, z, z, z, z, z, z) == 0
    # shape errors (the reference panics on empty variable lists, prove.rs:108)
    for m, n in ((0, 1), (1, 0), (-1, 2)):
        assert lib.gs_prove_batch(e.ctx, 0, n1, m, n, p, p, p, p, p, p, p, p, z, z, p, p) == 1  # GS_ERR_SHAPE
    assert lib.gs_prove_batch(e.ctx, 7, n1, 1, 1, p, p, p, p, p, p, p, p, z, z, p, p) in (1, 3)  # unknown type
    assert b"" != lib.gs_last_error(e.ctx)
    # null pointer for a required array
    assert lib.gs_prove_batch(e.ctx, 0, n1, 1, 1, z, p, p, p, p, p, p, p, z, z, p, p) == 3  # GS_ERR_ARG
    assert lib.gs_verify_batch(e.ctx, 0, n1, 1, 1, p, p, p, p, p, p, p, p, z) == 3
    # a context without a CRS refuses to prove / verify but still offers the CRS-free hooks
    fresh = gs.Engine(0, 0)
    assert lib.gs_prove_batch(fresh.ctx, 0, n1, 1, 1, p, p, p, p, p, p, p, p, z, z, p, p) == 4  # GS_ERR_NOCRS
    with pytest.raises(GsError):
        fresh.verify_batch(0, 1, 1, 1, one[:96], one[:192], one[:32], one[:576], one[:192], one[:384], one[:768],
                           one[:384])
    g1 = c.g1(c.golden["g1_smul"][0]["out"])
    out = fresh.g_mul_batch(1, g1.reshape(1, -1), c.fr(2).reshape(1, -1))
    assert c.g1_dec(out.reshape(-1).view(np.uint64)) == c.golden["g1_smul"][1]["out"]
    fresh.close()
    # bad curve / device ids
    h = ctypes.c_void_p()
    assert lib.gs_ctx_create(5, 0, ctypes.byref(h)) == 3
    assert lib.gs_ctx_create(0, 99, ctypes.byref(h)) == 2  # GS_ERR_DEVICE


@pytest.mark.parametrize("cname", CURVES)
def test_field_matrix_kats_on_the_prover_kernels(engines, cname):
    """The reference's Matrix<Fr> KATs (data_structures.rs:1726-1947) on the HIP path: gs_fr_matmul runs the product
    through k_prep_prove (Psi = R^T Gamma) -- and through the one-lane-per-output kernel for a shape past its switch.
      [[1,2,3],[4,5,6]] * [[7..10],[11..14],[15..18]] = [[74,80,86,92],[173,188,203,218]]
    plus a 1 x 1, a product with the identity, (r-1)-entries (wrap-around), and the C oracle on random matrices."""
    import os
    import sys

    from gsutil import REPO

    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import gs_ref_py as ref

    c, e = engines[cname]
    mat = lambda rows: [[c.fr(v) for v in r] for r in rows]
    dec = lambda out: [[c.fr_dec(v) for v in r] for r in out]
    assert dec(e.fr_matmul(mat([[1, 2, 3], [4, 5, 6]]), mat([[7, 8, 9, 10], [11, 12, 13, 14], [15, 16, 17, 18]]))) == \
        [[74, 80, 86, 92], [173, 188, 203, 218]]
    assert dec(e.fr_matmul(mat([[6]]), mat([[7]]))) == [[42]]
    assert dec(e.fr_matmul(mat([[1, 0], [0, 1]]), mat([[5, 6], [7, 8]]))) == [[5, 6], [7, 8]]
    r = c.r
    assert dec(e.fr_matmul(mat([[r - 1, r - 1]]), mat([[r - 1], [2]]))) == [[(1 - 2) % r]]
    rng = np.random.default_rng(5)
    for rows, inner, cols in ((2, 4, 4), (3, 5, 2), (2, 40, 33), (1, 1, 7)):  # 40 x 33 >= 1024: the wide kernel
        rnd = lambda a, b: [[int.from_bytes(rng.bytes(40), "little") % r for _ in range(b)] for _ in range(a)]
        A, B = rnd(rows, inner), rnd(inner, cols)
        got = dec(e.fr_matmul(mat(A), mat(B)))
        want = [[sum(A[i][k] * B[k][j] for k in range(inner)) % r for j in range(cols)] for i in range(rows)]
        assert got == want
        a = np.concatenate([c.fr(v) for row in A for v in row])
        b = np.concatenate([c.fr(v) for row in B for v in row])
        o = ref.fr_matmul(cname, rows, inner, cols, a, b).view(np.uint64).reshape(rows * cols, 4)
        assert [c.fr_dec(v) for v in o] == [v for row in want for v in row]


def test_contexts_share_crs_tables_and_rlc_rejects_zero_rho(engines):
    """Two contexts with the same CRS on one device share the 1.8 GB of window tables (gs_set_crs); a third context
    proving through the shared tables gives the same bytes.  gs_verify_batch_rlc refuses a zero exponent."""
    import torch

    import groth_sahai_rs_amd as gs
    from groth_sahai_rs_amd.capi import GsError

    c, e = engines["bls12_381"]
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    e2 = gs.Engine(0, 0)
    e2.set_crs(e._crs)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 256 << 20, "a second context with the same CRS must not build its own tables"
    case = c.golden["cases"][0]
    X, Y = enc_side(c, 0, "x", case["xvars"]), enc_side(c, 0, "y", case["yvars"])
    A, B = enc_side(c, 0, "x", case["a"]), enc_side(c, 0, "y", case["b"])
    G, R, S, T = (c.fr_mat(case[k]) for k in ("gamma", "R", "S", "T"))
    o1 = e.prove_batch(0, 1, case["m"], case["n"], X, Y, A, B, G, R, S, T)
    o2 = e2.prove_batch(0, 1, case["m"], case["n"], X, Y, A, B, G, R, S, T)
    assert all((o1[k] == o2[k]).all() for k in o1)
    tgt = enc_target(c, 0, case["target"])
    rho = np.array([3, 5, 0, 9], dtype=np.uint64)
    with pytest.raises(GsError) as ei:
        e2.verify_batch_rlc(0, 1, case["m"], case["n"], A, B, G, tgt, o2["xcoms"], o2["ycoms"], o2["pi"], o2["theta"], rho)
    assert ei.value.code == 3
    rho[2] = 7
    ok, _ = e2.verify_batch_rlc(0, 1, case["m"], case["n"], A, B, G, tgt, o2["xcoms"], o2["ycoms"], o2["pi"],
                                o2["theta"], rho)
    assert ok == 1
    e2.close()
    # the first context still works after the second one is gone (tables are reference-counted)
    o3 = e.prove_batch(0, 1, case["m"], case["n"], X, Y, A, B, G, R, S, T)
    assert all((o1[k] == o3[k]).all() for k in o1)
