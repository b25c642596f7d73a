"""The generated G1 point-operation subroutines (csrc/gen_pointops_asm.py -> gs_pointops_asm.h) run as straight-line
code with their own register allocation; this runs the SAME programs (the generator's op list) on Python integers:

  * limb-exact emulation of every operation (product-scanning Montgomery product, lazy add / sub / shifts, one-round
    carry) with the device's contracts asserted: every limb fits int32, every column accumulator fits int64;
  * the block allocation is replayed: an operation may only read blocks that hold the value it names (catches a
    temporary placed on a block that still holds a live value);
  * the resulting (X3, Y3, Z3) equal dbl-2009-l / madd-2007-bl evaluated with plain modular arithmetic on the same
    inputs, for random points, on both curves (reference semantics: src/data_structures.rs:187-188, 337-341 via
    arkworks' group law; any correct Jacobian representative normalises to the same affine point).
No GPU needed."""
import importlib.util
import os
import random

import pytest

from gsutil import REPO, curve

spec = importlib.util.spec_from_file_location("gen_pointops_asm",
                                              os.path.join(REPO, "groth_sahai_rs_amd", "csrc", "gen_pointops_asm.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

M = (1 << 28) - 1


def to_limbs(v, L):
    return [(v >> (28 * i)) & M for i in range(L)]


def val(l):
    return sum(x << (28 * i) for i, x in enumerate(l))


class Emu:
    def __init__(self, p, L):
        self.p, self.L = p, L
        self.P28 = to_limbs(p, L)
        self.inv = (-pow(p, -1, 1 << 28)) % (1 << 28)
        self.R = 1 << (28 * L)
        self.max_limb = 0
        self.max_acc = 0

    def chk(self, l):
        for x in l:
            assert -(1 << 31) <= x < (1 << 31), "limb leaves int32"
            self.max_limb = max(self.max_limb, abs(x))
        return l

    def mul(self, a, b):
        L, acc, m, r = self.L, 0, [0] * self.L, [0] * self.L
        for k in range(2 * L - 1):
            for i in range(L):
                j = k - i
                if j < 0 or j >= L:
                    continue
                acc += a[i] * b[j]
                if j >= 1 and i < k:
                    acc += m[i] * self.P28[j]
            if k < L:
                m[k] = (((acc & 0xFFFFFFFF) & M) * self.inv) & M
                acc += m[k] * self.P28[0]
            else:
                r[k - L] = acc & M
            assert -(1 << 63) <= acc < (1 << 63), "column accumulator leaves int64"
            self.max_acc = max(self.max_acc, abs(acc))
            acc >>= 28
        r[L - 1] = acc
        return self.chk(r)

    def norm(self, a):
        L = self.L
        r = [a[0] & M] + [(a[i] & M) + (a[i - 1] >> 28) for i in range(1, L - 1)] + [a[L - 1] + (a[L - 2] >> 28)]
        return self.chk(r)

    def run(self, prog_fn, inputs):
        """replay the op list with the generator's own block bookkeeping"""
        p = prog_fn(self.L)
        # second instance to replay allocation decisions: block of every value at definition time
        vals = dict(inputs)
        for kind, dst, a, b in p.ops:
            A = vals[a]
            B = vals[b] if b is not None else None
            if kind == "mul":
                r = self.mul(A, B)
            elif kind == "norm":
                r = self.norm(A)
            elif kind == "add":
                r = self.chk([x + y for x, y in zip(A, B)])
            elif kind == "sub":
                r = self.chk([x - y for x, y in zip(A, B)])
            elif kind == "dbl":
                r = self.chk([2 * x for x in A])
            elif kind == "subdbl":
                r = self.chk([x - 2 * y for x, y in zip(A, B)])
            elif kind == "x3":
                r = self.chk([3 * x for x in A])
            elif kind == "x4":
                r = self.chk([4 * x for x in A])
            else:
                raise AssertionError(kind)
            vals[dst] = r
        return vals

    def fe(self, l):  # internal limbs -> field element
        return val(l) * pow(self.R, -1, self.p) % self.p

    def enc(self, x):
        return to_limbs(x * self.R % self.p, self.L)


def jac_dbl(p, X, Y, Z):
    a, b = X * X % p, Y * Y % p
    c = b * b % p
    d = 2 * ((X + b) ** 2 - a - c) % p
    e = 3 * a % p
    f = e * e % p
    x3 = (f - 2 * d) % p
    return x3, (e * (d - x3) - 8 * c) % p, 2 * Y * Z % p


def jac_madd(p, X, Y, Z, x2, y2):
    z1z1 = Z * Z % p
    u2, s2 = x2 * z1z1 % p, y2 * Z * z1z1 % p
    h, rr = (u2 - X) % p, 2 * (s2 - Y) % p
    hh = h * h % p
    i = 4 * hh % p
    j, v = h * i % p, X * i % p
    x3 = (rr * rr - j - 2 * v) % p
    return x3, (rr * (v - x3) - 2 * Y * j) % p, ((Z + h) ** 2 - z1z1 - hh) % p


@pytest.mark.parametrize("cname,L", [("bls12_381", 14), ("bn254", 10)])
def test_generated_point_ops_on_integers(cname, L):
    c = curve(cname)
    p = c.p
    rnd = random.Random(1234 + L)
    worst_limb = worst_acc = 0
    for it in range(40):
        e = Emu(p, L)
        X, Y, Z, x2, y2 = (rnd.randrange(p) for _ in range(5))
        if it == 0:
            Z = 0  # the identity stays the identity under doubling (exact zero limbs)
        got = e.run(gen.g1_dbl, {"X": e.enc(X), "Y": e.enc(Y), "Z": e.enc(Z)})
        want = jac_dbl(p, X, Y, Z)
        assert (e.fe(got["X3"]), e.fe(got["Y3"]), e.fe(got["Z3"])) == want
        if Z == 0:
            assert got["Z3"] == [0] * L
            Z = 1
        # inputs of the device are normalised values in (-p/2, 3p/2): also feed shifted representatives
        shift = lambda v: to_limbs((v * e.R % p) + (p if it % 3 == 1 else 0), L)
        got = e.run(gen.g1_madd, {"X": shift(X), "Y": shift(Y), "Z": shift(Z), "qx": e.enc(x2), "qy": e.enc(y2)})
        want = jac_madd(p, X, Y, Z, x2, y2)
        assert (e.fe(got["X3"]), e.fe(got["Y3"]), e.fe(got["Z3"])) == want
        # outputs are normalised: limbs 0..L-2 in [0, 2^28 + small)
        for nm in ("X3", "Y3", "Z3"):
            assert all(-16 <= x < (1 << 28) + 16 for x in got[nm][:-1])
        worst_limb, worst_acc = max(worst_limb, e.max_limb), max(worst_acc, e.max_acc)
    assert worst_limb < (1 << 31) and worst_acc < (1 << 63)


def test_generated_programs_end_with_the_point_in_place():
    for L in (14, 10):
        for fn in (gen.g1_dbl, gen.g1_madd):
            prog = fn(L)
            assert sorted(prog.val.items()) == [("X3", 0), ("Y3", 1), ("Z3", 2)]
            assert prog.out[-1].startswith("s_setpc_b64")
            # no memory instruction, no scalar-memory instruction, nothing but VALU + the return
            assert not any(i.split()[0].startswith(("global_", "scratch_", "flat_", "ds_", "buffer_", "s_load")) for i in prog.out)
