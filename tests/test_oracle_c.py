"""The C restatement of the reference path (oracle/gs_ref.c) against the golden
fixtures of the big-integer oracle: both oracles must agree bit for bit on
every case, for both curves (CPU only)."""
import os
import sys

import numpy as np
import pytest

from gsutil import REPO, curve

sys.path.insert(0, os.path.join(REPO, "oracle"))
import gs_ref_py as ref  # noqa: E402

CURVES = ["bls12_381", "bn254"]


def crs_of(c):
    g = c.golden["crs"]
    return np.concatenate([c.com1(g["u"][0]), c.com1(g["u"][1]), c.com2(g["v"][0]), c.com2(g["v"][1]), c.g1(g["g1"]),
                           c.g2(g["g2"]), c.f12(g["gt"])])


def enc_side(c, ty, side, vals):
    xg, yg = ty in (0, 1), ty in (0, 2)
    if side == "x":
        return np.concatenate([c.g1(v) if xg else c.fr_hex(v) for v in vals])
    return np.concatenate([c.g2(v) if yg else c.fr_hex(v) for v in vals])


@pytest.mark.parametrize("cname", CURVES)
def test_sizes_and_arith(cname):
    c = curve(cname)
    FQ, FR, G1, G2, GT, CRS = ref.sizes(cname)
    assert (FQ, FR, G1, G2, GT) == (8 * c.nq, 32, 16 * c.nq, 32 * c.nq, 96 * c.nq)
    g = c.golden
    g1 = c.g1(g["g1_smul"][0]["out"])
    g2 = c.g2(g["g2_smul"][0]["out"])
    for e1, e2 in zip(g["g1_smul"], g["g2_smul"]):
        assert c.g1_dec(ref.g_mul(cname, 1, g1, c.fr_hex(e1["k"])).view(np.uint64)) == e1["out"]
        assert c.g2_dec(ref.g_mul(cname, 2, g2, c.fr_hex(e2["k"])).view(np.uint64)) == e2["out"]
    for pe in g["pairing"]:
        assert c.f12_dec(ref.multi_pairing(cname, 1, c.g1(pe["p"]), c.g2(pe["q"])).view(np.uint64)) == pe["out"]
    ps = g["pairing_sum"]
    out = ref.pairing_sum(cname, len(ps["x"]), np.concatenate([c.com1(v) for v in ps["x"]]),
                          np.concatenate([c.com2(v) for v in ps["y"]])).view(np.uint64).reshape(4, -1)
    assert [c.f12_dec(o) for o in out] == ps["out"]
    lm = g["left_mul"]
    rows, k = len(lm["lhs"]), len(lm["lhs"][0])
    o1 = ref.left_mul(cname, 1, rows, k, c.fr_mat(lm["lhs"]), np.concatenate([c.com1(v) for v in lm["com1"]]))
    o1 = o1.view(np.uint64).reshape(rows, 2, -1)
    assert [[c.g1_dec(v[0]), c.g1_dec(v[1])] for v in o1] == lm["out1"]
    o2 = ref.left_mul(cname, 2, rows, k, c.fr_mat(lm["lhs"]), np.concatenate([c.com2(v) for v in lm["com2"]]))
    o2 = o2.view(np.uint64).reshape(rows, 2, -1)
    assert [[c.g2_dec(v[0]), c.g2_dec(v[1])] for v in o2] == lm["out2"]


@pytest.mark.parametrize("cname", CURVES)
def test_cases(cname):
    c = curve(cname)
    crs = crs_of(c)
    for case in c.golden["cases"]:
        ty, m, n = case["type"], case["m"], case["n"]
        X, Y = enc_side(c, ty, "x", case["xvars"]), enc_side(c, ty, "y", case["yvars"])
        A, B = enc_side(c, ty, "x", case["a"]), enc_side(c, ty, "y", case["b"])
        G, R, S, T = (c.fr_mat(case[k]) for k in ("gamma", "R", "S", "T"))
        out = ref.commit_and_prove(cname, ty, m, n, X, Y, A, B, G, R, S, T, crs)
        xc = out["xcoms"].view(np.uint64).reshape(m, 2, -1)
        yc = out["ycoms"].view(np.uint64).reshape(n, 2, -1)
        pi = out["pi"].view(np.uint64).reshape(-1, 2, 4 * c.nq)
        th = out["theta"].view(np.uint64).reshape(-1, 2, 2 * c.nq)
        assert [[c.g1_dec(v[0]), c.g1_dec(v[1])] for v in xc] == case["xcoms"], case["name"]
        assert [[c.g2_dec(v[0]), c.g2_dec(v[1])] for v in yc] == case["ycoms"], case["name"]
        assert [[c.g2_dec(v[0]), c.g2_dec(v[1])] for v in pi] == case["pi"], case["name"]
        assert [[c.g1_dec(v[0]), c.g1_dec(v[1])] for v in th] == case["theta"], case["name"]
        if "verify" in case:
            tgt = {0: c.f12, 1: c.g1, 2: c.g2, 3: c.fr_hex}[ty](case["target"])
            assert ref.verify(cname, ty, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], out["pi"], out["theta"], crs) == 1
            bad = out["pi"].copy()
            bad[3] ^= 0x10
            assert ref.verify(cname, ty, m, n, A, B, G, tgt, out["xcoms"], out["ycoms"], bad, out["theta"], crs) == 0


def test_bench_entry_small():
    t, units, ok = ref.bench_ppe(2, 2, 2, 2)
    assert ok and units == 2 and t > 0


@pytest.mark.parametrize("cname", CURVES)
def test_field_matrix_kat(cname):
    """data_structures.rs:1726-1947: [[1,2,3],[4,5,6]] * [[7..10],[11..14],[15..18]] =
    [[74,80,86,92],[173,188,203,218]] (the Matrix<Fr> product behind R^T Gamma, Psi S, S^T Gamma^T)."""
    c = curve(cname)
    a = np.concatenate([c.fr(v) for v in (1, 2, 3, 4, 5, 6)])
    b = np.concatenate([c.fr(v) for v in range(7, 19)])
    out = ref.fr_matmul(cname, 2, 3, 4, a, b).view(np.uint64).reshape(8, 4)
    assert [c.fr_dec(o) for o in out] == [74, 80, 86, 92, 173, 188, 203, 218]
