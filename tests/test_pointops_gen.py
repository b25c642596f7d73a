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
spec2 = importlib.util.spec_from_file_location("gen_mul28_asm",
                                               os.path.join(REPO, "groth_sahai_rs_amd", "csrc", "gen_mul28_asm.py"))
mulgen = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(mulgen)

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


# ---- register-level interpretation of the EMITTED instructions (validates the allocator: G1 and G2 programs) ----------
import re


class Machine:
    """The dozen opcodes the generators emit, on Python integers, with the device's width contracts asserted: 32-bit
    lazy arithmetic must not wrap, a 64-bit accumulator must not leave int64."""

    def __init__(self, p, L):
        self.v, self.a, self.s = {}, {}, {}
        self.L = L
        self.vcc, self.sflag = False, {}
        pinv56 = pow(p, -1, 1 << 56)
        self.s[60], self.s[61] = pinv56 & 0xFFFFFFFF, pinv56 >> 32  # (the compact G2 addition's H = 0 filter)
        P28 = to_limbs(p, L)
        for i, x in enumerate(P28):
            self.s[40 + i] = x
        self.s[40 + L] = (-pow(p, -1, 1 << 28)) % (1 << 28)

    @staticmethod
    def s32(x):
        x &= 0xFFFFFFFF
        return x - (1 << 32) if x >> 31 else x

    def rd(self, tok):
        tok = tok.strip()
        if tok.startswith("v["):
            lo = int(tok[2:tok.index(":")])
            return (self.v[lo] & 0xFFFFFFFF) | ((self.v[lo + 1] & 0xFFFFFFFF) << 32)
        if tok.startswith("v"):
            return self.v[int(tok[1:])]
        if tok.startswith("s"):
            return self.s[int(tok[1:])]
        if tok.startswith("a"):
            return self.a[int(tok[1:])]
        return int(tok, 0)

    def wr64(self, tok, val):
        assert -(1 << 63) <= val < (1 << 63), "64-bit accumulator overflow"
        lo = int(tok[2:tok.index(":")])
        u = val & 0xFFFFFFFFFFFFFFFF
        self.v[lo], self.v[lo + 1] = self.s32(u), self.s32(u >> 32)

    def rd64s(self, tok):
        u = self.rd(tok)
        return u - (1 << 64) if u >> 63 else u

    def run(self, prog):
        prog = [x.strip() for x in prog]
        labels = {x[:-1]: i for i, x in enumerate(prog) if x.endswith(":")}
        pc = -1
        while pc + 1 < len(prog):
            pc += 1
            ins = prog[pc]
            if ins.startswith("s_setpc"):
                return
            if ins.startswith(".") or ins.startswith("s_mov_b64 s[36:37]"):
                continue
            if ins.startswith("s_cbranch_vccz"):
                if not self.vcc:
                    pc = labels[ins.split()[1]]
                continue
            if ins.startswith("s_mov_b64") or ins.startswith("s_and_b64"):
                op, rest = ins.split(None, 1)
                a = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)]
                val = lambda t: self.vcc if t == "vcc" else self.sflag[t]
                r = val(a[1]) if op == "s_mov_b64" else (val(a[1]) and val(a[2]))
                if a[0] == "vcc":
                    self.vcc = r
                else:
                    self.sflag[a[0]] = r
                continue
            if ins.startswith("s_swappc_b64"):
                # the compact G2 form calls the SHARED multiplier subroutines of gs_mul28_asm.h: run their generated
                # bodies, then forget what they are declared to destroy (a read of a destroyed register is a KeyError)
                L = self.L
                tgt = ins.split(",")[1].strip()
                if tgt == "s[56:57]":
                    self.run(mulgen.sub_body_fp2(L))
                    gone = list(range(6 * L, 7 * L + 4))
                else:
                    assert tgt == "s[58:59]", ins
                    self.run(mulgen.sub_body_fp2sqr(L))
                    gone = list(range(4 * L, 7 * L + 4))
                for k in gone:
                    self.v.pop(k, None)
                self.calls = getattr(self, "calls", 0) + 1
                continue
            op, rest = ins.split(None, 1)
            args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)]
            if op == "v_mad_i64_i32":
                d, _vcc, x, y, z = args
                acc = 0 if z == "0" else self.rd64s(z)
                self.wr64(d, self.s32(self.rd(x)) * self.s32(self.rd(y)) + acc)
            elif op == "v_mad_u64_u32":
                d, _vcc, x, y, z = args
                acc = self.rd64s(z)
                r = (self.rd(x) & 0xFFFFFFFF) * (self.rd(y) & 0xFFFFFFFF) + acc
                self.wr64(d, r)
            elif op == "v_mul_lo_u32":
                d, x, y = args
                self.v[int(d[1:])] = self.s32((self.rd(x) & 0xFFFFFFFF) * (self.rd(y) & 0xFFFFFFFF))
            elif op == "v_and_b32":
                d, x, y = args
                self.v[int(d[1:])] = self.s32((self.rd(x) & 0xFFFFFFFF) & (self.rd(y) & 0xFFFFFFFF))
            elif op == "v_ashrrev_i64":
                d, sh, x = args
                self.wr64(d, self.rd64s(x) >> int(sh))
            elif op == "v_ashrrev_i32":
                d, sh, x = args
                self.v[int(d[1:])] = self.s32(self.rd(x)) >> int(sh)
            elif op in ("v_mov_b32", "v_accvgpr_read_b32"):
                d, x = args
                self.v[int(d[1:])] = self.rd(x)
            elif op == "v_accvgpr_write_b32":
                d, x = args
                self.a[int(d[1:])] = self.rd(x)
            elif op in ("v_add_u32", "v_sub_u32"):
                d, x, y = args
                r = self.s32(self.rd(x)) + self.s32(self.rd(y)) if op == "v_add_u32" else self.s32(self.rd(x)) - self.s32(self.rd(y))
                assert -(1 << 31) <= r < (1 << 31), "lazy 32-bit arithmetic wrapped"
                self.v[int(d[1:])] = r
            elif op == "v_lshlrev_b32":
                d, sh, x = args
                r = self.s32(self.rd(x)) << int(sh)
                assert -(1 << 31) <= r < (1 << 31), "shift wrapped"
                self.v[int(d[1:])] = r
            elif op == "v_lshl_add_u32":
                d, x, sh, y = args
                r = (self.s32(self.rd(x)) << int(sh)) + self.s32(self.rd(y))
                assert -(1 << 31) <= r < (1 << 31)
                self.v[int(d[1:])] = r
            elif op == "s_mov_b32":
                d, x = args
                self.s[int(d[1:])] = int(x, 0)
            elif op == "v_bfe_i32":
                d, x, off, w = args
                val_ = (self.rd(x) & 0xFFFFFFFF) >> int(off) & ((1 << int(w)) - 1)
                self.v[int(d[1:])] = val_ - (1 << int(w)) if val_ >> (int(w) - 1) else val_
            elif op == "v_add3_u32":
                d, x, y, z = args
                self.v[int(d[1:])] = self.s32(self.rd(x) + self.rd(y) + self.rd(z))  # (wraps by design: low 64 bits of a product)
            elif op in ("v_add_co_u32", "v_addc_co_u32"):
                d, _vcc, x, y = args[:4]
                t = (self.rd(x) & 0xFFFFFFFF) + (self.rd(y) & 0xFFFFFFFF) + (int(self.vcc) if op == "v_addc_co_u32" else 0)
                self.vcc = bool(t >> 32)
                self.v[int(d[1:])] = self.s32(t)
            elif op == "v_cmp_eq_u32":
                _vcc, x, y = args
                self.vcc = (self.rd(x) & 0xFFFFFFFF) == (self.rd(y) & 0xFFFFFFFF)
            elif op == "v_cmp_ne_u32":
                _vcc, x, y = args
                self.vcc = (self.rd(x) & 0xFFFFFFFF) != (self.rd(y) & 0xFFFFFFFF)
            elif op == "v_cmp_ge_u32":
                _vcc, x, y = args
                self.vcc = (self.rd(x) & 0xFFFFFFFF) >= (self.rd(y) & 0xFFFFFFFF)
            else:
                raise AssertionError("opcode not modelled: " + ins)


def _put(mach, base, limbs):
    for i, x in enumerate(limbs):
        mach.v[base + i] = x


def _get(mach, base, L):
    return [mach.v[base + i] for i in range(L)]


@pytest.mark.parametrize("cname,L", [("bls12_381", 14), ("bn254", 10)])
def test_emitted_g1_instructions(cname, L):
    c = curve(cname)
    p = c.p
    rnd = random.Random(77 + L)
    B = gen.BASE
    for it in range(6):
        e = Emu(p, L)
        X, Y, Z, x2, y2 = (rnd.randrange(p) for _ in range(5))
        m = Machine(p, L)
        for k, val_ in enumerate((X, Y, Z)):
            _put(m, B + k * L, e.enc(val_))
        m.run(gen.g1_dbl(L).out)
        got = tuple(e.fe(_get(m, B + k * L, L)) for k in range(3))
        assert got == jac_dbl(p, X, Y, Z)
        top = B + gen.NB * L
        m = Machine(p, L)
        for k, val_ in enumerate((X, Y, Z, x2, y2)):
            _put(m, B + k * L, e.enc(val_))
        m.v[top + 2] = 0  # the caller's edge flag (T)
        m.run(gen.g1_madd(L).out)
        got = tuple(e.fe(_get(m, B + k * L, L)) for k in range(3))
        assert got == jac_madd(p, X, Y, Z, x2, y2)
        top = B + gen.NB * L
        assert m.v[top + 3] == 0  # flag: the generic formulas applied (a random H is never flagged "may be zero")
        # P = +-Q (H = 0 mod p): the subroutine returns at once, flag = 1, every operand register untouched
        x2e = X * pow(Z * Z % p, -1, p) % p
        m2 = Machine(p, L)
        for k, val_ in enumerate((X, Y, Z, x2e, y2)):
            _put(m2, B + k * L, e.enc(val_))
        m2.v[top + 2] = 0
        before = dict(m2.v)
        m2.run(gen.g1_madd(L).out)
        assert m2.v[top + 3] == 1
        assert all(m2.v[k] == v_ for k, v_ in before.items() if k != top + 2)  # (T, the edge flag, is a temporary after the entry test)
        # an identity operand flagged by the caller: immediate return, nothing touched, no multiplication executed
        m3 = Machine(p, L)
        for k, val_ in enumerate((X, Y, 0, x2, y2)):
            _put(m3, B + k * L, e.enc(val_))
        m3.v[top + 2] = 1
        before = dict(m3.v)
        m3.run(gen.g1_madd(L).out)
        assert m3.v[top + 3] == 1 and all(m3.v[k] == v_ for k, v_ in before.items()) and len(m3.v) == len(before) + 1


def f2mul(p, a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2(p, op, a, b=None):
    if op == "add":
        return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)
    if op == "sub":
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)
    if op == "k":
        return (a[0] * b % p, a[1] * b % p)


def jac2_dbl(p, X, Y, Z):
    a, b = f2mul(p, X, X), f2mul(p, Y, Y)
    c = f2mul(p, b, b)
    xb = f2(p, "add", X, b)
    d = f2(p, "k", f2(p, "sub", f2(p, "sub", f2mul(p, xb, xb), a), c), 2)
    e = f2(p, "k", a, 3)
    f = f2mul(p, e, e)
    x3 = f2(p, "sub", f, f2(p, "k", d, 2))
    y3 = f2(p, "sub", f2mul(p, e, f2(p, "sub", d, x3)), f2(p, "k", c, 8))
    return x3, y3, f2(p, "k", f2mul(p, Y, Z), 2)


def jac2_madd(p, X, Y, Z, x2, y2):
    z1z1 = f2mul(p, Z, Z)
    u2, s2 = f2mul(p, x2, z1z1), f2mul(p, f2mul(p, y2, Z), z1z1)
    h, rr = f2(p, "sub", u2, X), f2(p, "k", f2(p, "sub", s2, Y), 2)
    hh = f2mul(p, h, h)
    i = f2(p, "k", hh, 4)
    j, v = f2mul(p, h, i), f2mul(p, X, i)
    x3 = f2(p, "sub", f2(p, "sub", f2mul(p, rr, rr), j), f2(p, "k", v, 2))
    y3 = f2(p, "sub", f2mul(p, rr, f2(p, "sub", v, x3)), f2(p, "k", f2mul(p, Y, j), 2))
    zh = f2(p, "add", Z, h)
    return x3, y3, f2(p, "sub", f2(p, "sub", f2mul(p, zh, zh), z1z1), hh)


@pytest.mark.parametrize("form", ["straight", "compact"])
@pytest.mark.parametrize("cname,L", [("bls12_381", 14), ("bn254", 10)])
def test_emitted_g2_instructions(cname, L, form):
    """The G2 subroutines park values in AGPRs by a farthest-next-use allocator: run the emitted code itself.  `compact`
    is the form that calls the shared Fp2 multiplier subroutines (Prog2c)."""
    c = curve(cname)
    p = c.p
    rnd = random.Random(99 + L)
    cls = gen.Prog2c if form == "compact" else gen.Prog2
    for it in range(5):
        e = Emu(p, L)
        vals = [(rnd.randrange(p), rnd.randrange(p)) for _ in range(5)]
        if it == 0:
            vals[2] = (0, 0)  # the identity stays the identity under doubling
        for fn, nin, ref in ((gen.g2_dbl, 3, jac2_dbl), (gen.g2_madd, 5, jac2_madd)):
            prog = fn(L, cls)
            use = list(vals[:nin])
            if fn is gen.g2_madd and use[2] == (0, 0):
                use[2] = (1, 0)
            m = Machine(p, L)
            for k, (c0, c1) in enumerate(use):
                if form == "compact" and k >= 3:  # the addend arrives in AGPRs (after the parking blocks)
                    for q, cc in enumerate((c0, c1)):
                        for i, x in enumerate(e.enc(cc)):
                            m.a[prog.ain_base + (2 * (k - 3) + q) * L + i] = x
                    continue
                _put(m, prog.io_base + 2 * k * L, e.enc(c0))
                _put(m, prog.io_base + (2 * k + 1) * L, e.enc(c1))
            m.run(prog.out)
            got = tuple((e.fe(_get(m, prog.io_base + 2 * k * L, L)), e.fe(_get(m, prog.io_base + (2 * k + 1) * L, L)))
                        for k in range(3))
            assert got == ref(p, *use), (cname, fn.__name__, it)
            if form == "compact" and fn is gen.g2_madd:
                assert m.v[int(prog.hout[0][1:])] == 0  # flag: the generic formulas applied
                # P = +-Q (H = 0 mod p): the subroutine returns at once, flag = 1, operands untouched
                X, Y, Z, x2, y2 = use
                zz = f2mul(p, Z, Z)
                zinv = pow((zz[0] * zz[0] + zz[1] * zz[1]) % p, -1, p)
                x2e = f2mul(p, X, (zz[0] * zinv % p, -zz[1] * zinv % p))  # x2 = X / Z^2
                m2 = Machine(p, L)
                for k, (c0, c1) in enumerate((X, Y, Z)):
                    _put(m2, prog.io_base + 2 * k * L, e.enc(c0))
                    _put(m2, prog.io_base + (2 * k + 1) * L, e.enc(c1))
                for k, (c0, c1) in enumerate((x2e, y2)):
                    for q, cc in enumerate((c0, c1)):
                        for i, x in enumerate(e.enc(cc)):
                            m2.a[prog.ain_base + (2 * k + q) * L + i] = x
                before_v = {k: v_ for k, v_ in m2.v.items()}
                before_a = dict(m2.a)
                m2.run(prog.out)
                assert m2.v[int(prog.hout[0][1:])] == 1
                assert all(m2.v[k] == v_ for k, v_ in before_v.items())
                assert all(m2.a[k] == v_ for k, v_ in before_a.items())
            if fn is gen.g2_dbl and use[2] == (0, 0):
                assert _get(m, prog.io_base + 4 * L, L) == [0] * L and _get(m, prog.io_base + 5 * L, L) == [0] * L
            # nothing outside the declared registers was written
            hi = max(k for k in m.v)
            assert hi < prog.top, hi
            assert all(k < Prog2_NPARK_L(prog) + (4 * L if form == "compact" else 0) for k in m.a)


def Prog2_NPARK_L(prog):
    return prog.NPARK * prog.L


def f6mul_ref(p, xi, a, b):
    """(a0 + a1 v + a2 v^2)(b0 + b1 v + b2 v^2) with v^3 = xi, coefficients in Fp2 = Fp[u]/(u^2 + 1)"""
    def add(x, y):
        return ((x[0] + y[0]) % p, (x[1] + y[1]) % p)
    mx = lambda x: f2mul(p, x, xi)
    m = lambda i, j: f2mul(p, a[i], b[j])
    r0 = add(m(0, 0), mx(add(m(1, 2), m(2, 1))))
    r1 = add(add(m(0, 1), m(1, 0)), mx(m(2, 2)))
    r2 = add(add(m(0, 2), m(1, 1)), m(2, 0))
    return r0, r1, r2


@pytest.mark.parametrize("cname,L,xi", [("bls12_381", 14, (1, 1)), ("bn254", 10, (9, 1))])
def test_emitted_f6_mul(cname, L, xi):
    """The general Fp6 product as a generated subroutine (six nested calls of the shared Fp2 product): a in/out in VGPR
    blocks, b in AGPR blocks; the emitted instructions against plain modular arithmetic."""
    c = curve(cname)
    p = c.p
    rnd = random.Random(5 + L)
    prog = gen.f6_mul(L)
    for it in range(6):
        e = Emu(p, L)
        a = [(rnd.randrange(p), rnd.randrange(p)) for _ in range(3)]
        b = [(rnd.randrange(p), rnd.randrange(p)) for _ in range(3)]
        if it == 0:
            a[1], b[2] = (0, 0), (p - 1, 0)
        m = Machine(p, L)
        for k in range(3):
            for q in range(2):
                _put(m, prog.io_base + (2 * k + q) * L, e.enc(a[k][q]))
                for i, x in enumerate(e.enc(b[k][q])):
                    m.a[prog.ain_base + (2 * k + q) * L + i] = x
        m.run(prog.out)
        got = tuple((e.fe(_get(m, prog.io_base + 2 * k * L, L)), e.fe(_get(m, prog.io_base + (2 * k + 1) * L, L)))
                    for k in range(3))
        assert got == f6mul_ref(p, xi, a, b), (cname, it)
        assert m.calls == 6
        assert max(m.v) < prog.top and max(m.a) < (prog.NPARK + 6) * L
        # outputs are weakly normalised limbs (the contract every consumer of an Fp6 product relies on)
        for k in range(6):
            limbs = _get(m, prog.io_base + k * L, L)
            assert all(0 <= x < (1 << 28) + 64 for x in limbs[:-1]), limbs


def test_register_pairs_are_even_aligned():
    """gfx950 takes 64-bit VGPR operands only on even-aligned pairs -- an assembler error, but one that shows up only
    when the whole library is built (the holders' inline asm is not assembled by a -S probe)."""
    progs = []
    for L in (14, 10):
        progs += [gen.g1_dbl(L), gen.g1_madd(L), gen.g2_dbl(L, gen.Prog2c), gen.g2_madd(L, gen.Prog2c), gen.g2_dbl(L),
                  gen.g2_madd(L), gen.f6_mul(L)]
    for prog in progs:
        for ins in prog.out:
            for mm in re.finditer(r"v\[(\d+):(\d+)\]", ins):
                assert int(mm.group(1)) % 2 == 0 and int(mm.group(2)) == int(mm.group(1)) + 1, ins
