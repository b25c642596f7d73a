"""TEST INFRASTRUCTURE ONLY (see oracle/gs_oracle.py's header): big-integer restatement of the
canonical wire format ark-serialize ^0.5 gives the reference's derives
(src/data_structures.rs:128,132; src/prover/commit.rs:18,24; src/prover/prove.rs:55;
src/statement.rs:61-97,117-179; src/generator.rs:35).

PARITY UNPINNED at the byte level: /root/reference holds no serialised fixture (its tests are
round trips, data_structures.rs:1270-1310, commit.rs:300-340, prove.rs:600-640, statement.rs:215-390)
and arkworks cannot run here.  The encodings below restate the published formats [ark-mem]:
  * ark-serialize: PrimeField = canonical integer, little-endian, ceil(bits/8) bytes; extension
    fields coefficient by coefficient (c0 first); Vec<T> = u64-LE length then the items; structs
    field by field in declaration order; u8 = 1 byte.
  * ark-bls12-381 (>= 0.4) overrides point encoding with the zcash format: big-endian x (G2: c1
    then c0), top three bits of the first byte = compressed / infinity / y-lexicographically-largest.
    Pinned here against the well-known compressed encodings of the standard generators (0x97f1d3..,
    0x93e02b..), which those rules reproduce.
  * ark-ec default short-Weierstrass encoding (BN254): little-endian x (G2: c0 then c1) with SWFlags
    in the top two bits of the last byte: 0x80 = y > -y, 0x40 = infinity; uncompressed appends y and
    carries the flags on y.
Values are the oracle's: Fp ints, Fp2 = (c0, c1), points = (x, y) or None, Fp12 nested tuples."""
import struct

import gs_oracle as O  # noqa: E402  (oracle/ is put on sys.path by the tests)


def _fq_len():
    return (O.P.bit_length() + 7) // 8


def _fr_len():
    return (O.R.bit_length() + 7) // 8


def enc_fr(a):
    return (a % O.R).to_bytes(_fr_len(), "little")


def dec_fr(b):
    v = int.from_bytes(b, "little")
    if v >= O.R:
        raise ValueError("non-canonical Fr")
    return v


def enc_gt(f):
    return b"".join((c % O.P).to_bytes(_fq_len(), "little") for c in O.f12_flat(f))


def dec_gt(b, validate=True):
    n = _fq_len()
    vals = [int.from_bytes(b[i * n:(i + 1) * n], "little") for i in range(12)]
    if any(v >= O.P for v in vals):
        raise ValueError("non-canonical Fq")
    f = O.f12_unflat(vals)
    if validate and O.f12_pow(f, O.R) != O.f12_unflat([1] + [0] * 11):
        raise ValueError("GT element not in the r-torsion")
    return f


def _largest_fq(y):
    return y > (O.P - 1) // 2


def _largest(y):
    if isinstance(y, tuple):
        return _largest_fq(y[1]) if y[1] != 0 else _largest_fq(y[0])
    return _largest_fq(y)


def _coords(v):
    return list(v) if isinstance(v, tuple) else [v]


def enc_point(pt, group, compressed):
    """group 1 / 2; pt = None is the identity."""
    n = _fq_len()
    nc = 1 if group == 1 else 2
    total = (nc if compressed else 2 * nc) * n
    zcash = O.C.name == "bls12_381"
    if pt is None:
        out = bytearray(total)
        if zcash:
            out[0] |= (0x80 if compressed else 0) | 0x40
        else:
            out[-1] |= 0x40
        return bytes(out)
    x, y = pt
    vals = _coords(x) + ([] if compressed else _coords(y))
    if zcash:
        order = (list(reversed(_coords(x))) + ([] if compressed else list(reversed(_coords(y)))))
        out = bytearray(b"".join(v.to_bytes(n, "big") for v in order))
        out[0] |= (0x80 if compressed else 0) | (0x20 if compressed and _largest(y) else 0)
    else:
        out = bytearray(b"".join(v.to_bytes(n, "little") for v in vals))
        out[-1] |= 0x80 if _largest(y) else 0
    return bytes(out)


def _sqrt_fq(a):
    r = pow(a % O.P, (O.P + 1) // 4, O.P)
    return r if r * r % O.P == a % O.P else None


def _sqrt_f2(a):
    # brute-force-free: a^((p^2+7)/16)-style methods need p^2 = 9 mod 16 cases; use the norm method
    a0, a1 = a[0] % O.P, a[1] % O.P
    if a1 == 0:
        s = _sqrt_fq(a0)
        if s is not None:
            return (s, 0)
        s = _sqrt_fq(-a0 % O.P)
        return None if s is None else (0, s)
    n = _sqrt_fq((a0 * a0 + a1 * a1) % O.P)
    if n is None:
        return None
    inv2 = pow(2, -1, O.P)
    for d in ((a0 + n) * inv2 % O.P, (a0 - n) * inv2 % O.P):
        c0 = _sqrt_fq(d)
        if c0 is not None and c0 != 0:
            c1 = a1 * pow(2 * c0, -1, O.P) % O.P
            if O.f2_sqr((c0, c1)) == (a0, a1):
                return (c0, c1)
    return None


def dec_point(b, group, compressed, validate=True):
    n = _fq_len()
    nc = 1 if group == 1 else 2
    total = (nc if compressed else 2 * nc) * n
    if len(b) != total:
        raise ValueError("length")
    zcash = O.C.name == "bls12_381"
    buf = bytearray(b)
    if zcash:
        fb = buf[0]
        if bool(fb & 0x80) != compressed:
            raise ValueError("compression flag")
        inf, largest = bool(fb & 0x40), bool(fb & 0x20)
        if not compressed and largest:
            raise ValueError("sort flag on an uncompressed point")
        if inf and largest:
            raise ValueError("sort flag on the identity")  # ark-bls12-381 EncodingFlags::get_flags
        buf[0] &= 0x1F
        vals = [int.from_bytes(buf[i * n:(i + 1) * n], "big") for i in range(total // n)]
        xs = list(reversed(vals[:nc]))
        ys = list(reversed(vals[nc:]))
    else:
        fb = buf[-1]
        inf, largest = bool(fb & 0x40), bool(fb & 0x80)
        if inf and largest:
            raise ValueError("flags")
        buf[-1] &= 0x3F
        vals = [int.from_bytes(buf[i * n:(i + 1) * n], "little") for i in range(total // n)]
        xs, ys = vals[:nc], vals[nc:]
    if any(v >= O.P for v in vals):
        raise ValueError("non-canonical coordinate")
    if inf:
        if any(vals):
            raise ValueError("identity with non-zero coordinates")
        return None
    x = xs[0] if nc == 1 else tuple(xs)
    if nc == 1:
        rhs = (x ** 3 + O.C.b) % O.P
    else:
        rhs = O.f2_add(O.f2_mul(O.f2_sqr(x), x), O.C.b2)
    if compressed:
        y = _sqrt_fq(rhs) if nc == 1 else _sqrt_f2(rhs)
        if y is None:
            raise ValueError("x is not on the curve")
        if _largest(y) != largest:
            y = (-y) % O.P if nc == 1 else O.f2_neg(y)
    else:
        y = ys[0] if nc == 1 else tuple(ys)
        if (y * y % O.P if nc == 1 else O.f2_sqr(y)) != rhs:
            raise ValueError("not on the curve")
    pt = (x, y)
    if validate:
        if O.ec_mul(O.FP if nc == 1 else O.FP2, O.R, pt) is not None:  # not g*_mul: those reduce the scalar mod r
            raise ValueError("not in the prime-order subgroup")
    return pt


# ---- struct framing ---------------------------------------------------------------------------------

def enc_vec(items, enc):
    return struct.pack("<Q", len(items)) + b"".join(enc(x) for x in items)


def enc_matrix_fr(m):
    return enc_vec(m, lambda row: enc_vec(row, enc_fr))


def enc_com(c, group, compressed):
    return enc_point(c[0], group, compressed) + enc_point(c[1], group, compressed)


def enc_commit(coms, rand, group, compressed):  # Commit1 / Commit2 {coms, rand}
    return enc_vec(coms, lambda c: enc_com(c, group, compressed)) + enc_matrix_fr(rand)


def enc_equ_proof(pi, theta, equ_type, rand, compressed):  # EquProof {pi, theta, equ_type, rand}
    return (enc_vec(pi, lambda c: enc_com(c, 2, compressed)) + enc_vec(theta, lambda c: enc_com(c, 1, compressed))
            + bytes([equ_type]) + enc_matrix_fr(rand))


def enc_crs(crs, compressed):  # CRS {u, v, g1_gen, g2_gen, gt_gen}
    return (enc_vec(crs["u"], lambda c: enc_com(c, 1, compressed)) + enc_vec(crs["v"], lambda c: enc_com(c, 2, compressed))
            + enc_point(crs["g1"], 1, compressed) + enc_point(crs["g2"], 2, compressed) + enc_gt(crs["gt"]))


def enc_equation(ty, a, b, gamma, target, compressed):  # PPE / MSMEG1 / MSMEG2 / QuadEqu
    ea = (lambda v: enc_point(v, 1, compressed)) if ty in (0, 1) else enc_fr
    eb = (lambda v: enc_point(v, 2, compressed)) if ty in (0, 2) else enc_fr
    et = {0: enc_gt, 1: lambda v: enc_point(v, 1, compressed), 2: lambda v: enc_point(v, 2, compressed), 3: enc_fr}[ty]
    return enc_vec(a, ea) + enc_vec(b, eb) + enc_matrix_fr(gamma) + et(target)
