#!/usr/bin/env python3
"""Big-integer oracle for the Groth-Sahai prove/verify hot path (BLS12-381, BN254).

TEST INFRASTRUCTURE ONLY.  Nothing under groth_sahai_rs_amd/ may import this
file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
and only as the checker.

What it restates (file:line are relative to /root/reference):
  * Com1/Com2/ComT algebra, iota maps, pairing / pairing_sum
        src/data_structures.rs:181-251, 300-388, 399-541
  * Matrix<Fr> / Matrix<Com> products (left_mul / right_mul)
        src/data_structures.rs:587-742, 768-913
  * commit_G1 / batch_commit_G1 / ... scalar_to_B1/B2   src/prover/commit.rs:59-256
  * Provable::{prove, commit_and_prove} for PPE/MSMEG1/MSMEG2/QuadEqu
        src/prover/prove.rs:71-489
  * Verifiable::verify for the four equation types       src/verifier.rs:23-157
  * CRS shape (binding key)                              src/generator.rs:81-118

The arithmetic the reference delegates to arkworks (ark-ec / ark-ff ^0.5,
ark-bls12-381 ^0.5 -- NOT present under /root/reference, no Cargo.lock) is
restated here from the published algorithms with plain Python integers and the
most literal textbook methods available (affine chord-and-tangent, Miller loop
on the untwisted point in Fp12, final exponentiation by a plain square-and-
multiply with the integer exponent), so that it shares no code and no clever
trick with either the C restatement (oracle/gs_ref.c) or the HIP kernels.

PARITY STATUS: "parity unpinned" at the byte level.  The reference holds no
golden vector, hex constant or serialized fixture for this path (SURVEY.md
section 8c) and neither Rust nor arkworks can run in this container.  This
oracle pins itself with textbook identities (on-curve, group order,
bilinearity, non-degeneracy, e(P,Q)^r = 1, the reference's own algebraic test
properties) -- see selfcheck().  The one convention that cannot be checked here
is arkworks' pairing exponent: ark-ec's BLS12 final exponentiation raises the
Miller value to (p^6-1)(p^2+1) * [(x-1)^2 (x+p)(x^2+p^2-1) + 3], i.e. the CUBE
of the textbook reduced pairing (eprint 2020/875); for BN254 it raises the hard
part to 2x(6x^2+3x+1) * (p^4-p^2+1)/r (Fuentes-Castaneda et al.).  Both are
switches below (Curve.fe_cofactor).
"""
import hashlib
import json
import sys

# ----------------------------------------------------------------------------
# Curve parameter sets
# ----------------------------------------------------------------------------


class Curve:
    pass


def _bls12_381():
    c = Curve()
    c.name = "bls12_381"
    c.curve_id = 0
    c.p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    c.r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    c.x = -0xD201000000010000
    c.fq_limbs = 6  # u64 limbs of Fq
    c.fr_limbs = 4
    c.b = 4  # y^2 = x^3 + 4
    c.xi = (1, 1)  # Fp6 = Fp2[v]/(v^3 - xi), xi = 1 + u
    c.twist = "M"  # E': y^2 = x^3 + b*xi
    c.b2 = (4, 4)
    c.g1 = (
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    )
    c.g2 = (
        (
            0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
        ),
        (
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
        ),
    )
    # Miller loop: f_{|x|,Q}(P), inverted because x < 0 (ate pairing, t-1 = x)
    c.miller_count = abs(c.x)
    c.miller_neg = True
    c.bn_frobenius_lines = False
    # arkworks: hard part exponent = 3*(p^4-p^2+1)/r  [ark-mem, eprint 2020/875]
    c.fe_cofactor = 3
    return c


def _bn254():
    c = Curve()
    c.name = "bn254"
    c.curve_id = 1
    c.x = 4965661367192848881
    x = c.x
    c.p = 36 * x**4 + 36 * x**3 + 24 * x**2 + 6 * x + 1
    c.r = 36 * x**4 + 36 * x**3 + 18 * x**2 + 6 * x + 1
    c.fq_limbs = 4
    c.fr_limbs = 4
    c.b = 3
    c.xi = (9, 1)
    c.twist = "D"  # E': y^2 = x^3 + b/xi
    c.b2 = None  # filled below
    c.g1 = (1, 2)
    c.g2 = (
        (
            10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
        ),
        (
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531,
        ),
    )
    c.miller_count = 6 * x + 2
    c.miller_neg = False
    c.bn_frobenius_lines = True
    # arkworks BN: hard part yields f^{2x(6x^2+3x+1)*(p^4-p^2+1)/r}  [ark-mem]
    c.fe_cofactor = 2 * x * (6 * x * x + 3 * x + 1)
    return c


# ----------------------------------------------------------------------------
# Field towers, written against a module-level "current curve" for brevity.
# ----------------------------------------------------------------------------

C = None  # current curve
P = None
R = None
XI = None


def set_curve(c):
    global C, P, R, XI
    C, P, R, XI = c, c.p, c.r, c.xi
    if c.b2 is None:
        c.b2 = f2_mul((c.b, 0), f2_inv(c.xi))
    return c


def fp_inv(a):
    return pow(a % P, -1, P)


# --- Fp2 = Fp[u]/(u^2+1) ------------------------------------------------------
F2_0 = (0, 0)
F2_1 = (1, 0)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    n = fp_inv(a[0] * a[0] + a[1] * a[1])
    return (a[0] * n % P, (-a[1]) * n % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


# --- Fp6 = Fp2[v]/(v^3 - xi) -------------------------------------------------
F6_0 = (F2_0, F2_0, F2_0)
F6_1 = (F2_1, F2_0, F2_0)


def f6_add(a, b):
    return tuple(f2_add(x, y) for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple(f2_sub(x, y) for x, y in zip(a, b))


def f6_neg(a):
    return tuple(f2_neg(x) for x in a)


def f6_mul(a, b):
    # schoolbook, v^3 = xi
    t = [F2_0] * 5
    for i in range(3):
        for j in range(3):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return (
        f2_add(t[0], f2_mul(XI, t[3])),
        f2_add(t[1], f2_mul(XI, t[4])),
        t[2],
    )


def f6_mul_v(a):
    # (a0 + a1 v + a2 v^2) * v = xi*a2 + a0 v + a1 v^2
    return (f2_mul(XI, a[2]), a[0], a[1])


def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul(XI, f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul(XI, f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    n = f2_add(f2_mul(a0, t0), f2_mul(XI, f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    ni = f2_inv(n)
    return (f2_mul(t0, ni), f2_mul(t1, ni), f2_mul(t2, ni))


# --- Fp12 = Fp6[w]/(w^2 - v) --------------------------------------------------
F12_0 = (F6_0, F6_0)
F12_1 = (F6_1, F6_0)


def f12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    t0 = f6_mul(a0, b0)
    t1 = f6_mul(a1, b1)
    c0 = f6_add(t0, f6_mul_v(t1))
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a0, a1), f6_add(b0, b1)), t0), t1)
    return (c0, c1)


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_inv(a):
    a0, a1 = a
    n = f6_sub(f6_mul(a0, a0), f6_mul_v(f6_mul(a1, a1)))
    ni = f6_inv(n)
    return (f6_mul(a0, ni), f6_neg(f6_mul(a1, ni)))


def f12_pow(a, e):
    if e < 0:
        return f12_pow(f12_inv(a), -e)
    r = F12_1
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


def f12_from_fp(a):
    return (((a % P, 0), F2_0, F2_0), F6_0)


def f12_from_f2(a):
    return ((a, F2_0, F2_0), F6_0)


def f12_flat(a):
    """12 Fp coefficients in arkworks order c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1."""
    out = []
    for h in a:
        for q in h:
            out += [q[0], q[1]]
    return out


def f12_unflat(v):
    it = iter(v)
    return tuple(tuple((next(it), next(it)) for _ in range(3)) for _ in range(2))


F12_W = (F6_0, F6_1)  # w


# ----------------------------------------------------------------------------
# Elliptic-curve groups (affine, identity = None)
# ----------------------------------------------------------------------------


class Fld:
    """Minimal field vtable so one affine group law serves Fp, Fp2 and Fp12."""

    def __init__(self, add, sub, mul, inv, neg, zero, eq=None):
        self.add, self.sub, self.mul, self.inv, self.neg, self.zero = add, sub, mul, inv, neg, zero


FP = Fld(
    lambda a, b: (a + b) % P,
    lambda a, b: (a - b) % P,
    lambda a, b: a * b % P,
    fp_inv,
    lambda a: (-a) % P,
    0,
)
FP2 = Fld(f2_add, f2_sub, f2_mul, f2_inv, f2_neg, F2_0)
FP12 = Fld(
    lambda a, b: (f6_add(a[0], b[0]), f6_add(a[1], b[1])),
    lambda a, b: (f6_sub(a[0], b[0]), f6_sub(a[1], b[1])),
    f12_mul,
    f12_inv,
    lambda a: (f6_neg(a[0]), f6_neg(a[1])),
    F12_0,
)


def ec_add(F, p1, p2):
    """Affine chord-and-tangent on y^2 = x^3 + b (a = 0)."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 != y2 or y1 == F.zero:
            return None
        three_x2 = F.mul(F.add(F.add(x1, x1), x1), x1)
        lam = F.mul(three_x2, F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F, p):
    return None if p is None else (p[0], F.neg(p[1]))


def ec_mul(F, k, p):
    """Plain double-and-add, scalar taken as a non-negative integer."""
    if k < 0:
        return ec_mul(F, -k, ec_neg(F, p))
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = ec_add(F, acc, acc)
        if bit == "1":
            acc = ec_add(F, acc, p)
    return acc


def g1_add(a, b):
    return ec_add(FP, a, b)


def g1_neg(a):
    return ec_neg(FP, a)


def g1_mul(k, a):
    return ec_mul(FP, k % R, a)


def g2_add(a, b):
    return ec_add(FP2, a, b)


def g2_neg(a):
    return ec_neg(FP2, a)


def g2_mul(k, a):
    return ec_mul(FP2, k % R, a)


def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - C.b) % P == 0


def g2_on_curve(q):
    if q is None:
        return True
    x, y = q
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), C.b2)) == F2_0


# ----------------------------------------------------------------------------
# Pairing: textbook Miller loop on the UNTWISTED point, in Fp12
# ----------------------------------------------------------------------------


def untwist(q):
    """E'(Fp2) -> E(Fp12).  w^6 = xi.
    M-type (y^2 = x^3 + b*xi):  (x', y') -> (x'/w^2, y'/w^3)
    D-type (y^2 = x^3 + b/xi):  (x', y') -> (x'*w^2, y'*w^3)"""
    x, y = q
    w2 = f12_mul(F12_W, F12_W)
    w3 = f12_mul(w2, F12_W)
    if C.twist == "M":
        return (f12_mul(f12_from_f2(x), f12_inv(w2)), f12_mul(f12_from_f2(y), f12_inv(w3)))
    return (f12_mul(f12_from_f2(x), w2), f12_mul(f12_from_f2(y), w3))


def _line(T, S, Pt):
    """Value at Pt of the line through T and S (tangent if T == S); vertical
    lines are omitted (they lie in a proper subfield and die in the final exp)."""
    F = FP12
    x1, y1 = T
    x2, y2 = S
    xp, yp = Pt
    if x1 == x2 and y1 == y2:
        lam = F.mul(F.mul(F.add(F.add(x1, x1), x1), x1), F.inv(F.add(y1, y1)))
    elif x1 == x2:
        return F.sub(xp, x1)  # vertical (only hit by degenerate inputs)
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    return F.sub(F.sub(yp, y1), F.mul(lam, F.sub(xp, x1)))


def frob_fp12(a, k=1):
    """a^(p^k) by plain exponentiation of the basis: slow but literal."""
    return f12_pow(a, P**k)


def miller(p, q):
    """f_{s,Q}(P) with s = curve loop count; P in E(Fp), Q in E'(Fp2)."""
    if p is None or q is None:
        return F12_1
    Pt = (f12_from_fp(p[0]), f12_from_fp(p[1]))
    Q = untwist(q)
    T = Q
    f = F12_1
    for bit in bin(C.miller_count)[3:]:
        f = f12_mul(f12_sqr(f), _line(T, T, Pt))
        T = ec_add(FP12, T, T)
        if bit == "1":
            f = f12_mul(f, _line(T, Q, Pt))
            T = ec_add(FP12, T, Q)
    if C.bn_frobenius_lines:
        # optimal ate on BN: two more line additions with pi(Q), -pi^2(Q)
        Q1 = (frob_fp12(Q[0]), frob_fp12(Q[1]))
        Q2 = (frob_fp12(Q[0], 2), frob_fp12(Q[1], 2))
        Q2 = ec_neg(FP12, Q2)
        f = f12_mul(f, _line(T, Q1, Pt))
        T = ec_add(FP12, T, Q1)
        f = f12_mul(f, _line(T, Q2, Pt))
    if C.miller_neg:
        f = f12_inv(f)
    return f


_FE_CACHE = {}


def final_exp(f):
    key = C.name
    if key not in _FE_CACHE:
        hard = (P**4 - P**2 + 1) // R
        assert (P**4 - P**2 + 1) % R == 0
        if C.name == "bls12_381":
            x = C.x
            assert (x - 1) ** 2 * (x + P) * (x * x + P * P - 1) + 3 == 3 * hard
        _FE_CACHE[key] = hard * C.fe_cofactor
    e = _FE_CACHE[key]
    f1 = f12_mul(f12_conj(f), f12_inv(f))  # ^(p^6 - 1)
    f2 = f12_mul(f12_pow(f1, P * P), f1)  # ^(p^2 + 1)
    return f12_pow(f2, e)


def pairing(p, q):
    return final_exp(miller(p, q))


def multi_pairing(ps, qs):
    """E::multi_pairing: product of Miller values, ONE final exponentiation;
    pairs with an identity argument are skipped [ark-mem]."""
    f = F12_1
    for p, q in zip(ps, qs):
        if p is None or q is None:
            continue
        f = f12_mul(f, miller(p, q))
    return final_exp(f)


# ----------------------------------------------------------------------------
# GS commitment-group algebra (src/data_structures.rs)
# ----------------------------------------------------------------------------
# Com1 = (G1, G1) tuple, Com2 = (G2, G2) tuple, ComT = 4-tuple of Fp12 (00,01,10,11).


def com1_add(a, b):  # data_structures.rs:181-190
    return (g1_add(a[0], b[0]), g1_add(a[1], b[1]))


def com2_add(a, b):
    return (g2_add(a[0], b[0]), g2_add(a[1], b[1]))


def com1_neg(a):
    return (g1_neg(a[0]), g1_neg(a[1]))


def com2_neg(a):
    return (g2_neg(a[0]), g2_neg(a[1]))


def com1_smul(a, s):  # data_structures.rs:336-342
    return (g1_mul(s, a[0]), g1_mul(s, a[1]))


def com2_smul(a, s):  # data_structures.rs:381-387
    return (g2_mul(s, a[0]), g2_mul(s, a[1]))


COM1_ZERO = (None, None)
COM2_ZERO = (None, None)


def lin1(x):  # iota_1, data_structures.rs:310-312
    return (None, x)


def lin2(y):  # iota_2, data_structures.rs:355-357
    return (None, y)


def slin1(x, crs):  # iota_1', data_structures.rs:323-326
    return com1_smul(com1_add(crs["u"][1], lin1(crs["g1"])), x)


def slin2(y, crs):  # iota_2', data_structures.rs:368-371
    return com2_smul(com2_add(crs["v"][1], lin2(crs["g2"])), y)


def comt_pairing(x, y):  # data_structures.rs:484-491
    return (pairing(x[0], y[0]), pairing(x[0], y[1]), pairing(x[1], y[0]), pairing(x[1], y[1]))


def comt_pairing_sum(xs, ys):  # data_structures.rs:494-502
    assert len(xs) == len(ys)
    return tuple(
        multi_pairing([x[a] for x in xs], [y[b] for y in ys]) for a in (0, 1) for b in (0, 1)
    )


def comt_add(a, b):  # GT "addition" is Fp12 multiplication, data_structures.rs:399-410
    return tuple(f12_mul(x, y) for x, y in zip(a, b))


def comt_lin_ppe(t):  # data_structures.rs:509-516
    return (F12_1, F12_1, F12_1, t)


def comt_lin_msmeg1(t, crs):  # :519-524
    return comt_pairing(lin1(t), slin2(1, crs))


def comt_lin_msmeg2(t, crs):  # :527-532
    return comt_pairing(slin1(1, crs), lin2(t))


def comt_lin_quad(t, crs):  # :535-540
    return comt_pairing(slin1(1, crs), com2_smul(slin2(1, crs), t))


# --- matrices (lists of rows) -------------------------------------------------


def transpose(m):
    return [list(r) for r in zip(*m)]


def fr_matmul(a, b):  # Matrix<Fr> right_mul, data_structures.rs:824-869
    if not a or not a[0] or not b or not b[0]:
        return []
    assert len(a[0]) == len(b)
    return [[sum(a[i][k] * b[k][j] for k in range(len(b))) % R for j in range(len(b[0]))] for i in range(len(a))]


def fr_matadd(a, b):
    assert len(a) == len(b) and len(a[0]) == len(b[0])
    return [[(x + y) % R for x, y in zip(ra, rb)] for ra, rb in zip(a, b)]


def fr_matneg(a):
    return [[(-x) % R for x in r] for r in a]


def com_left_mul(colvec, lhs, smul, add, zero):
    """Matrix<Com>::left_mul for a (k x 1) column of Com elements and an (r x k)
    Fr matrix: out[i] = sum_k lhs[i][k] * col[k]   (data_structures.rs:696-742)."""
    if not lhs or not lhs[0] or not colvec:
        return []
    assert len(lhs[0]) == len(colvec)
    out = []
    for row in lhs:
        acc = zero
        for k, c in enumerate(colvec):
            acc = add(acc, smul(c, row[k]))
        out.append(acc)
    return out


def com1_left_mul(col, lhs):
    return com_left_mul(col, lhs, com1_smul, com1_add, COM1_ZERO)


def com2_left_mul(col, lhs):
    return com_left_mul(col, lhs, com2_smul, com2_add, COM2_ZERO)


# ----------------------------------------------------------------------------
# CRS of the reference's shape (src/generator.rs:81-118), scalars supplied
# ----------------------------------------------------------------------------


def make_crs(p1, p2, a1, a2, t1, t2):
    q1, q2 = g1_mul(a1, p1), g2_mul(a2, p2)
    u1, u2 = g1_mul(t1, p1), g2_mul(t2, p2)
    v1, v2 = g1_mul(t1, q1), g2_mul(t2, q2)  # binding key: generator.rs:57-58
    return {
        "u": [(p1, q1), (u1, v1)],
        "v": [(p2, q2), (u2, v2)],
        "g1": p1,
        "g2": p2,
        "gt": pairing(p1, p2),
    }


# ----------------------------------------------------------------------------
# Commit (src/prover/commit.rs); randomness is always an input
# ----------------------------------------------------------------------------


def batch_commit_g1(xs, crs, Rm):  # commit.rs:78-100 ; Rm is m x 2
    ru = com1_left_mul(crs["u"], Rm)
    return [com1_add(lin1(x), c) for x, c in zip(xs, ru)]


def batch_commit_g2(ys, crs, Sm):  # commit.rs:178-200
    sv = com2_left_mul(crs["v"], Sm)
    return [com2_add(lin2(y), c) for y, c in zip(ys, sv)]


def batch_commit_scalar_b1(xs, crs, r):  # commit.rs:125-156 ; r is m' x 1
    return [com1_add(slin1(x, crs), com1_smul(crs["u"][0], ri[0])) for x, ri in zip(xs, r)]


def batch_commit_scalar_b2(ys, crs, s):  # commit.rs:225-256
    return [com2_add(slin2(y, crs), com2_smul(crs["v"][0], si[0])) for y, si in zip(ys, s)]


# ----------------------------------------------------------------------------
# Prove (src/prover/prove.rs).  equ = dict(type, a, b, gamma, target)
# ----------------------------------------------------------------------------
PPE, MSMEG1, MSMEG2, QUAD = 0, 1, 2, 3


def prove(equ, xvars, yvars, Rm, Sm, T, crs):
    """Literal restatement of Provable::prove for all four types
    (prove.rs:92-171, 195-274, 298-379, 409-488).  Rm/Sm are the commit
    randomness matrices, T the proof randomness in the reference's shape."""
    ty = equ["type"]
    gamma = equ["gamma"]
    assert len(xvars) == len(Rm) and len(gamma) == len(Rm)
    assert len(yvars) == len(Sm) and len(gamma[0]) == len(Sm)
    x_is_g = ty in (PPE, MSMEG1)
    y_is_g = ty in (PPE, MSMEG2)
    assert len(Rm[0]) == (2 if x_is_g else 1)
    assert len(Sm[0]) == (2 if y_is_g else 1)
    assert len(T) == len(Sm[0]) and len(T[0]) == len(Rm[0])

    Rt, St = transpose(Rm), transpose(Sm)
    map_b = [lin2(b) for b in equ["b"]] if y_is_g else [slin2(b, crs) for b in equ["b"]]
    map_y = [lin2(y) for y in yvars] if y_is_g else [slin2(y, crs) for y in yvars]
    map_a = [lin1(a) for a in equ["a"]] if x_is_g else [slin1(a, crs) for a in equ["a"]]
    map_x = [lin1(x) for x in xvars] if x_is_g else [slin1(x, crs) for x in xvars]

    x_rand_lin_b = com2_left_mul(map_b, Rt)
    x_rand_stmt = fr_matmul(Rt, gamma)
    x_rand_stmt_lin_y = com2_left_mul(map_y, x_rand_stmt)
    pf_rand_stmt = fr_matadd(fr_matmul(fr_matmul(Rt, gamma), Sm), fr_matneg(transpose(T)))
    vcol = crs["v"] if y_is_g else [crs["v"][0]]
    pf_rand_stmt_com2 = com2_left_mul(vcol, pf_rand_stmt)
    pi = [com2_add(com2_add(a, b), c) for a, b, c in zip(x_rand_lin_b, x_rand_stmt_lin_y, pf_rand_stmt_com2)]

    y_rand_lin_a = com1_left_mul(map_a, St)
    y_rand_stmt = fr_matmul(St, transpose(gamma))
    y_rand_stmt_lin_x = com1_left_mul(map_x, y_rand_stmt)
    ucol = crs["u"] if x_is_g else [crs["u"][0]]
    pf_rand_com1 = com1_left_mul(ucol, T)
    theta = [com1_add(com1_add(a, b), c) for a, b, c in zip(y_rand_lin_a, y_rand_stmt_lin_x, pf_rand_com1)]
    assert len(pi) == len(Rm[0]) and len(theta) == len(Sm[0])
    return pi, theta


def commit_and_prove(equ, xvars, yvars, Rm, Sm, T, crs):
    """prove.rs:72-90 etc.; draw order in the reference is R, then S, then T."""
    ty = equ["type"]
    xc = batch_commit_g1(xvars, crs, Rm) if ty in (PPE, MSMEG1) else batch_commit_scalar_b1(xvars, crs, Rm)
    yc = batch_commit_g2(yvars, crs, Sm) if ty in (PPE, MSMEG2) else batch_commit_scalar_b2(yvars, crs, Sm)
    pi, theta = prove(equ, xvars, yvars, Rm, Sm, T, crs)
    return xc, yc, pi, theta


# ----------------------------------------------------------------------------
# Verify (src/verifier.rs:23-157), literal: five pairing_sums, 4 FEs each
# ----------------------------------------------------------------------------


def verify(equ, xc, yc, pi, theta, crs):
    ty = equ["type"]
    map_a = [lin1(a) for a in equ["a"]] if ty in (PPE, MSMEG1) else [slin1(a, crs) for a in equ["a"]]
    map_b = [lin2(b) for b in equ["b"]] if ty in (PPE, MSMEG2) else [slin2(b, crs) for b in equ["b"]]
    lin_a_com_y = comt_pairing_sum(map_a, yc)
    com_x_lin_b = comt_pairing_sum(xc, map_b)
    stmt_com_y = com2_left_mul(yc, equ["gamma"])
    com_x_stmt_com_y = comt_pairing_sum(xc, stmt_com_y)
    if ty == PPE:
        lin_t = comt_lin_ppe(equ["target"])
    elif ty == MSMEG1:
        lin_t = comt_lin_msmeg1(equ["target"], crs)
    elif ty == MSMEG2:
        lin_t = comt_lin_msmeg2(equ["target"], crs)
    else:
        lin_t = comt_lin_quad(equ["target"], crs)
    if ty in (PPE, MSMEG1):
        com1_pf2 = comt_pairing_sum(crs["u"], pi)
    else:
        com1_pf2 = comt_pairing(crs["u"][0], pi[0])
    if ty in (PPE, MSMEG2):
        pf1_com2 = comt_pairing_sum(theta, crs["v"])
    else:
        pf1_com2 = comt_pairing(theta[0], crs["v"][0])
    lhs = comt_add(comt_add(lin_a_com_y, com_x_lin_b), com_x_stmt_com_y)
    rhs = comt_add(comt_add(lin_t, com1_pf2), pf1_com2)
    return lhs == rhs


# ----------------------------------------------------------------------------
# Deterministic test PRNG (splitmix64), shared by oracle, C restatement, bench
# ----------------------------------------------------------------------------


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        """Uniform-ish scalar: 4 limbs (LE), top bits masked, rejection-sampled
        as a canonical integer < r."""
        bits = R.bit_length()
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << bits) - 1
            if v < R:
                return v


# ----------------------------------------------------------------------------
# Self-check: the identities that pin this oracle (see module docstring)
# ----------------------------------------------------------------------------


def selfcheck(verbose=True):
    def say(*a):
        if verbose:
            print(*a)
            sys.stdout.flush()

    g1, g2 = C.g1, C.g2
    assert g1_on_curve(g1) and g2_on_curve(g2)
    assert g1_mul(R - 1, g1) == g1_neg(g1) and ec_mul(FP, R, g1) is None
    assert ec_mul(FP2, R, g2) is None
    say("  generators on curve, order r: ok")
    # untwist lands on E(Fp12)
    Q = untwist(g2)
    y2 = f12_mul(Q[1], Q[1])
    x3 = f12_mul(f12_mul(Q[0], Q[0]), Q[0])
    assert FP12.sub(y2, x3) == f12_from_fp(C.b)
    say("  untwist(g2) on E(Fp12): ok")
    e = pairing(g1, g2)
    assert e != F12_1, "degenerate"
    assert f12_pow(e, R) == F12_1
    a, b = 0x1234567, 0x7654321AB
    assert pairing(g1_mul(a, g1), g2_mul(b, g2)) == f12_pow(e, a * b)
    assert multi_pairing([g1_mul(a, g1), g1_neg(g1)], [g2, g2_mul(a, g2)]) == F12_1
    assert pairing(None, g2) == F12_1 and pairing(g1, None) == F12_1
    say("  pairing bilinear, non-degenerate, order r: ok")
    # exponent convention: e == (textbook reduced ate)^cofactor
    hard = (P**4 - P**2 + 1) // R
    m = miller(g1, g2)
    f1 = f12_mul(f12_conj(m), f12_inv(m))
    f2 = f12_mul(f12_pow(f1, P * P), f1)
    assert f12_pow(f12_pow(f2, hard), C.fe_cofactor) == e
    say("  exponent convention = textbook^%d: ok" % C.fe_cofactor if C.fe_cofactor < 10 else "  exponent convention: ok")
    return e


# ----------------------------------------------------------------------------
# Hex (de)serialisation helpers for fixtures: canonical integers, NOT Montgomery
# ----------------------------------------------------------------------------


def hx(v):
    return "%x" % v


def enc_g1(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def enc_g2(q):
    return None if q is None else [hx(q[0][0]), hx(q[0][1]), hx(q[1][0]), hx(q[1][1])]


def enc_f12(f):
    return [hx(v) for v in f12_flat(f)]


def dec_g1(v):
    return None if v is None else (int(v[0], 16), int(v[1], 16))


def dec_g2(v):
    return None if v is None else ((int(v[0], 16), int(v[1], 16)), (int(v[2], 16), int(v[3], 16)))


def dec_f12(v):
    return f12_unflat([int(s, 16) for s in v])


BLS12_381 = _bls12_381()
BN254 = _bn254()
set_curve(BLS12_381)

if __name__ == "__main__":
    for c in (BLS12_381, BN254):
        if len(sys.argv) > 1 and c.name not in sys.argv[1:]:
            continue
        set_curve(c)
        print("selfcheck", c.name)
        selfcheck()
    set_curve(BLS12_381)
