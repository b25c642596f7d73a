"""Subgroup safety at the boundary.

The reference's `Com::scalar_mul` is plain double-and-add on ANY curve point (src/data_structures.rs:336-342, :381-387);
the engine's variable-base scalar multiplications use the GLV / psi-GLS endomorphisms, which act as a scalar only on the
r-torsion.  Two ways to close the gap are tested here:

  * gs_validate_points: the wire decoder's tests (canonical coordinates, on the curve or the identity flag, r-torsion)
    on in-memory limbs -- subgroup points, the identity, a curve point OUTSIDE the subgroup, a point of the COFACTOR
    subgroup ([r]P), an off-curve point and a non-canonical coordinate, both groups, both curves (BN254 G1 has
    cofactor 1: every curve point is in the subgroup there);
  * gs_set_option("endo", 0): every variable-base scalar multiplication runs as a plain signed-window double-and-add
    lane -- scalar multiples of points outside the subgroup and whole PROOFS over such points then equal the C oracle's
    plain arithmetic (oracle/gs_ref.c) bit for bit.  (With the endomorphisms on, the same inputs are outside the
    contract: include/gs_amd.h.)
The verifier is not part of this: a pairing of points outside the r-torsion is not bilinear, so neither the reference's
verdict nor anyone else's means anything there."""
import os
import sys

import numpy as np
import pytest

from gsutil import REPO, curve

sys.path.insert(0, os.path.join(REPO, "oracle"))

pytestmark = pytest.mark.gpu

B1 = {"bls12_381": 4, "bn254": 3}


def _sqrt_fp(a, p):
    a %= p
    y = pow(a, (p + 1) // 4, p)  # p = 3 mod 4 on both curves
    return y if y * y % p == a else None


def _sqrt_fp2(a, p):
    a0, a1 = a[0] % p, a[1] % p
    if a1 == 0:
        y = _sqrt_fp(a0, p)
        if y is not None:
            return (y, 0)
        y = _sqrt_fp(-a0, p)
        return (0, y)
    s = _sqrt_fp(a0 * a0 + a1 * a1, p)
    if s is None:
        return None
    inv2 = pow(2, -1, p)
    for t in ((a0 + s) * inv2 % p, (a0 - s) * inv2 % p):
        x0 = _sqrt_fp(t, p)
        if x0:
            x1 = a1 * pow(2 * x0, -1, p) % p
            if ((x0 * x0 - x1 * x1) % p, 2 * x0 * x1 % p) == (a0, a1):
                return (x0, x1)
    return None


def curve_points(cname, group, count, seed):
    """`count` points of the curve (G1) / the twist (G2) found by trying x = seed, seed + 1, ...: for BLS12-381 (and on
    the BN254 twist) such a point lies outside the prime-order subgroup with overwhelming probability."""
    import gs_oracle as orc

    c = curve(cname)
    p = c.p
    orc.set_curve(orc._bls12_381() if cname == "bls12_381" else orc._bn254())
    out, x = [], seed
    while len(out) < count:
        x += 1
        if group == 1:
            y = _sqrt_fp(x * x * x + B1[cname], p)
            if y:
                assert orc.g1_on_curve((x, y))
                out.append((x, y))
        else:
            xx = (x, 2 * x + 1)
            rhs = orc.f2_add(orc.f2_mul(orc.f2_sqr(xx), xx), orc.C.b2)
            y = _sqrt_fp2(rhs, p)
            if y:
                assert orc.g2_on_curve((xx, y))
                out.append((xx, y))
    return out


def limbs(c, pt, group):
    if group == 1:
        return np.concatenate([c.fq(pt[0]), c.fq(pt[1])])
    return np.concatenate([c.fq(pt[0][0]), c.fq(pt[0][1]), c.fq(pt[1][0]), c.fq(pt[1][1])])


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
@pytest.mark.parametrize("group", [1, 2])
def test_validate_points(cname, cid, group):
    import groth_sahai_rs_amd as gs
    import gs_oracle as orc
    import gs_ref_py as ref

    c = curve(cname)
    eng = gs.Engine(cid, 0)
    gen = (c.g1 if group == 1 else c.g2)(c.golden["crs"]["g1" if group == 1 else "g2"])
    rng = np.random.default_rng(31 + group)
    rnd_fr = lambda: c.fr(int.from_bytes(rng.bytes(40), "little") % c.r)
    good = [ref.g_mul(cname, group, gen, rnd_fr()).view(np.uint64) for _ in range(3)]
    outside = curve_points(cname, group, 2, 1000 * group + cid)
    # [r]P: a point of the cofactor subgroup (order divides h): on the curve, not in the r-torsion (ec_mul is the
    # oracle's plain double-and-add; g1_mul / g2_mul reduce the scalar mod r first)
    cof = orc.ec_mul(orc.FP if group == 1 else orc.FP2, c.r, outside[0])
    off = limbs(c, outside[1], group).copy()
    off[-1] ^= np.uint64(1 << 7)  # y disturbed: off the curve
    noncanon = good[0].copy()
    v = sum(int(w) << (64 * i) for i, w in enumerate(noncanon[:c.nq])) + c.p  # the same x plus p: not < p any more
    assert v < (1 << (64 * c.nq))
    noncanon[:c.nq] = [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(c.nq)]
    ident = np.zeros_like(good[0])
    pts = good + [ident, limbs(c, outside[0], group), limbs(c, outside[1], group), off, noncanon]
    in_sub = cname == "bn254" and group == 1  # cofactor 1: every curve point is in the subgroup
    want = [1, 1, 1, 1, int(in_sub), int(in_sub), 0, 0]
    if cof is not None:
        pts.append(limbs(c, cof, group))
        want.append(0)
    else:
        assert in_sub  # [r]P = O only when P was in the subgroup
    ok = eng.validate_points(group, np.concatenate(pts))
    assert ok.tolist() == want, (cname, group, ok.tolist(), want)
    eng.close()


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
def test_plain_scalar_multiplication_on_points_outside_the_subgroup(cname, cid):
    import groth_sahai_rs_amd as gs
    import gs_ref_py as ref

    c = curve(cname)
    eng = gs.Engine(cid, 0)
    eng.set_option("endo", 0)
    rng = np.random.default_rng(5 + cid)
    fr = lambda: c.fr(int.from_bytes(rng.bytes(40), "little") % c.r)
    for group in (1, 2):
        pts = [limbs(c, p, group) for p in curve_points(cname, group, 5, 4000 * group + cid)]
        ks = [fr() for _ in pts] + [c.fr(0), c.fr(1), c.fr(c.r - 1)]
        pts = pts + pts[:3]
        P, K = np.concatenate(pts), np.concatenate(ks)
        got = eng.g_mul_batch(group, P, K)
        per = got.size // len(pts)
        for i, (p, k) in enumerate(zip(pts, ks)):
            want = ref.g_mul(cname, group, p, k)
            assert (got.view(np.uint8).reshape(-1)[i * per:(i + 1) * per] == want).all(), (cname, group, i)
    eng.close()


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("bn254", 1)])
@pytest.mark.parametrize("ty", [0, 1, 2])
def test_proofs_over_points_outside_the_subgroup_match_the_oracle(cname, cid, ty):
    """commit_and_prove with EVERY group element (variables and constants) a curve point outside the r-torsion, the
    endomorphisms off: commitments, pi and theta equal the C restatement of the reference's plain arithmetic."""
    import groth_sahai_rs_amd as gs
    import gs_ref_py as ref

    c = curve(cname)
    eng = gs.Engine(cid, 0)
    g = c.golden["crs"]
    crs = np.concatenate([c.com1(g["u"][0]), c.com1(g["u"][1]), c.com2(g["v"][0]), c.com2(g["v"][1]),
                          c.g1(g["g1"]), c.g2(g["g2"]), c.f12(g["gt"])])
    eng.set_crs(crs)
    eng.set_option("endo", 0)
    N, m, n = 5, 3, 2
    xg, yg = ty in (0, 1), ty in (0, 2)
    kx, ky = (2 if xg else 1), (2 if yg else 1)
    rng = np.random.default_rng(900 + ty)
    fr = lambda k: np.concatenate([c.fr(int.from_bytes(rng.bytes(40), "little") % c.r) for _ in range(k)])
    g1s = iter(curve_points(cname, 1, 2 * N * (m + n), 7000 + ty))
    g2s = iter(curve_points(cname, 2, 2 * N * (m + n), 9000 + ty))
    g1 = lambda k: np.concatenate([limbs(c, next(g1s), 1) for _ in range(k)])
    g2 = lambda k: np.concatenate([limbs(c, next(g2s), 2) for _ in range(k)])
    X = g1(N * m) if xg else fr(N * m)
    A = g1(N * n) if xg else fr(N * n)
    Y = g2(N * n) if yg else fr(N * n)
    B = g2(N * m) if yg else fr(N * m)
    G, R, S, T = fr(N * m * n), fr(N * m * kx), fr(N * n * ky), fr(N * ky * kx)
    out = eng.prove_batch(ty, N, m, n, X, Y, A, B, G, R, S, T)
    u8 = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    cut = lambda a, e: u8(a)[e * (u8(a).size // N):(e + 1) * (u8(a).size // N)]
    for e in range(N):
        want = ref.commit_and_prove(cname, ty, m, n, cut(X, e), cut(Y, e), cut(A, e), cut(B, e), cut(G, e), cut(R, e),
                                    cut(S, e), cut(T, e), crs)
        for k in ("xcoms", "ycoms", "pi", "theta"):
            assert (cut(out[k], e) == want[k]).all(), (cname, ty, e, k)
    eng.close()
