#!/usr/bin/env python3
"""Generate gs_pointops_asm.h: whole G1 point operations (Jacobian doubling, mixed addition) as ONE gfx950 subroutine
each, with their own register allocation.

Why (round 4, profiles/r4/diag_2p16.json): the Straus kernels spend 13-21 % of their wave cycles parked in s_waitcnt
and ~25 % of their instructions outside the multiplier.  Built from C++ with the multiplier behind fixed-register asm
statements, every Fq product marshals 28 dwords in and 14 out (v_mov), hipcc keeps two coordinates of the running point
in the private segment across the doubling run (26 dword loads + 26 stores per point operation, each batch one exposed
memory latency at one wave per SIMD), and any such wait also waits for the table entry that was requested ahead.  A G1
point operation fits the register file (14 blocks of L registers), so here it is straight-line code: the
product-scanning multiplier is instantiated per call site ON the registers its operands already live in -- no operand
moves, no spills, no memory instruction at all -- and the lazy additions / carry rounds are the ones of jac_dbl /
jac_madd in gs_curve.cuh, statement by statement (same limb-growth contract, which tests/test_pointops_gen.py re-checks
by running the generated programs on integers).

Register map of a subroutine (L = limbs: 14 BLS12-381, 10 BN254):
    block k = v[B+kL .. B+kL+L-1], k = 0..10, B = 48;  ACC = v[B+11L : B+11L+1], T = v[B+11L+2], HOUT = v[B+11L+3], v[B+11L+4]
    in / out:  X = block 0, Y = block 1, Z = block 2 (the running Jacobian point, updated IN PLACE)
    madd only: qx = block 3, qy = block 4 (affine addend, destroyed); HOUT = limbs 0 and 1 of the normalised
               H = U2 - X1 (the caller's cheap "H may be 0 mod p" filter: gs_fq28.cuh maybe_zero_limbs01)
    modulus in s40.. as for the multiplier subroutines, return address in s[34:35].
The edge cases of the addition (either operand the identity, H = 0) are the CALLER's: it tests before / after the call
and takes the C++ path on a saved copy (gs_curve.cuh, jac_madd_fast).
"""
import os

M28 = "0xfffffff"
NB = 11  # value blocks: a mixed addition has at most 10 values live, block NB - 1 is the squaring's doubled operand.
# (11 and not more: v48 .. v(48 + 11 L + 4) leaves the compiler 97 VGPRs beside a BLS12-381 subroutine, which is what
# lets the G1 Straus kernels be built for TWO waves per SIMD -- 256 registers in all)
# first register of block 0.  The multiplier subroutines (gs_mul28_asm.h) own v0..v44; with the point operations on the
# same registers every call of one had to move the running point out of the other's way (84 moves per point operation
# in the first build).  From v48 up the running point simply STAYS in blocks 0..2 from one call to the next.
BASE = 48


class Prog:
    def __init__(self, L, name):
        self.L, self.name = L, name
        self.out = []        # instructions
        self.ops = []        # IR for the integer emulation: (kind, dst, a, b)
        self.val = {}        # value name -> block
        self.free_blocks = list(range(NB - 1))  # block NB-1 is the squaring's doubled operand
        top = BASE + NB * L
        self.acc = "v[%d:%d]" % (top, top + 1)
        self.lo = "v%d" % top
        self.t = "v%d" % (top + 2)
        self.hout = "v%d" % (top + 3)
        self.hout1 = "v%d" % (top + 4)
        self.peak = 0

    # ---- registers
    def r(self, blk, i):
        return "v%d" % (BASE + blk * self.L + i)

    def P(self, i):
        return "s%d" % (40 + i)

    @property
    def INV(self):
        return "s%d" % (40 + self.L)

    # ---- values
    def pin(self, name, blk):
        assert blk in self.free_blocks, (name, blk)
        self.free_blocks.remove(blk)
        self.val[name] = blk

    def new(self, name, into=None):
        assert name not in self.val, name
        if into is not None:
            assert into in self.free_blocks, "block %d not free for %s (live: %s)" % (into, name, self.val)
            blk = into
        else:
            # blocks 0..2 hold the running point and receive its new coordinates (`into`): temporaries stay off them
            cand = [k for k in self.free_blocks if k >= 3]
            assert cand, "out of blocks at %s (live: %s)" % (name, self.val)
            blk = cand[0]
        self.free_blocks.remove(blk)
        self.val[name] = blk
        self.peak = max(self.peak, NB - 1 - len(self.free_blocks))
        return blk

    def free(self, *names):
        for n in names:
            blk = self.val.pop(n)
            self.free_blocks.append(blk)
            self.free_blocks.sort()

    def b(self, name):
        assert name in self.val, "value %s is not live" % name
        return self.val[name]

    # ---- operations (dst is a NEW value unless inplace is given)
    def mul(self, dst, a, b, into=None):
        A, B = self.b(a), self.b(b)
        D = self.new(dst, into)
        assert D not in (A, B)
        self._mulbody(D, A, B, square=False)
        self.ops.append(("mul", dst, a, b))

    def sqr(self, dst, a, into=None):
        A = self.b(a)
        D = self.new(dst, into)
        assert D != A
        self._mulbody(D, A, NB - 1, square=True)
        self.ops.append(("mul", dst, a, a))

    def _mulbody(self, D, A, B, square):
        L, o = self.L, self.out
        # every instruction of the multiplier is 8 bytes long and wants to start 8-byte aligned (gen_mul28_asm.py: the
        # same code at 4 mod 8 ran 10 % slower); the linear sections before it are a mix of 4- and 8-byte encodings
        o.append(".p2align 3")
        if square:  # doubled operand in the scratch block
            for i in range(L):
                o.append("v_lshlrev_b32 %s, 1, %s" % (self.r(B, i), self.r(A, i)))
        first = True
        for k in range(2 * L - 1):
            for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
                j = k - i
                if square and i > j:
                    continue
                y = (self.r(A, j) if i == j else self.r(B, j)) if square else self.r(B, j)
                o.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (self.acc, self.r(A, i), y, "0" if first else self.acc))
                first = False
            for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (self.acc, self.r(D, i), self.P(k - i), self.acc))
            if k < L:
                o.append("v_mul_lo_u32 %s, %s, %s" % (self.r(D, k), self.lo, self.INV))
                o.append("v_and_b32 %s, %s, %s" % (self.r(D, k), M28, self.r(D, k)))
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (self.acc, self.r(D, k), self.P(0), self.acc))
            else:
                o.append("v_and_b32 %s, %s, %s" % (self.r(D, k - L), M28, self.lo))
            o.append("v_ashrrev_i64 %s, 28, %s" % (self.acc, self.acc))
        o.append("v_mov_b32 %s, %s" % (self.r(D, L - 1), self.lo))

    def lin(self, kind, dst, a, b=None, into=None, inplace=False):
        """add / sub / dbl; inplace: dst replaces a (same block)"""
        A = self.b(a)
        Bk = self.b(b) if b is not None else None
        if inplace:
            D = A
            self.val[dst] = self.val.pop(a)
        else:
            D = self.new(dst, into)
        for i in range(self.L):
            if kind == "add":
                self.out.append("v_add_u32 %s, %s, %s" % (self.r(D, i), self.r(A, i), self.r(Bk, i)))
            elif kind == "sub":
                self.out.append("v_sub_u32 %s, %s, %s" % (self.r(D, i), self.r(A, i), self.r(Bk, i)))
            elif kind == "dbl":
                self.out.append("v_lshlrev_b32 %s, 1, %s" % (self.r(D, i), self.r(A, i)))
            elif kind == "subdbl":  # a - 2 b
                self.out.append("v_lshlrev_b32 %s, 1, %s" % (self.t, self.r(Bk, i)))
                self.out.append("v_sub_u32 %s, %s, %s" % (self.r(D, i), self.r(A, i), self.t))
            elif kind == "x3":  # 3 a = (a << 1) + a
                self.out.append("v_lshl_add_u32 %s, %s, 1, %s" % (self.r(D, i), self.r(A, i), self.r(A, i)))
            elif kind == "x4":
                self.out.append("v_lshlrev_b32 %s, 2, %s" % (self.r(D, i), self.r(A, i)))
            else:
                raise ValueError(kind)
        self.ops.append((kind, dst, a, b))

    def norm(self, name):
        """one parallel carry round, IN PLACE (top limb first: limb i needs the old limb i - 1)"""
        D, L = self.b(name), self.L
        for i in range(L - 1, 0, -1):
            self.out.append("v_ashrrev_i32 %s, 28, %s" % (self.t, self.r(D, i - 1)))
            if i != L - 1:
                self.out.append("v_and_b32 %s, %s, %s" % (self.r(D, i), M28, self.r(D, i)))
            self.out.append("v_add_u32 %s, %s, %s" % (self.r(D, i), self.r(D, i), self.t))
        self.out.append("v_and_b32 %s, %s, %s" % (self.r(D, 0), M28, self.r(D, 0)))
        self.ops.append(("norm", name, name, None))

    def export_limb0(self, name):
        self.out.append("v_mov_b32 %s, %s" % (self.hout, self.r(self.b(name), 0)))
        self.out.append("v_mov_b32 %s, %s" % (self.hout1, self.r(self.b(name), 1)))

    def ret(self):
        self.out.append("s_setpc_b64 s[34:35]")


def g1_dbl(L):
    """jac_dbl (dbl-2009-l, gs_curve.cuh), in place on blocks 0..2"""
    p = Prog(L, "dbl")
    p.pin("X", 0), p.pin("Y", 1), p.pin("Z", 2)
    p.sqr("a", "X")
    p.sqr("b", "Y")
    p.sqr("c", "b")
    p.lin("add", "xb", "X", "b")           # A = 2 (sqr_l2 contract)
    p.free("X", "b")
    p.sqr("t2", "xb")
    p.free("xb")
    p.lin("sub", "d", "t2", "a", inplace=True)
    p.lin("sub", "d1", "d", "c", inplace=True)
    p.lin("dbl", "d2", "d1", inplace=True)  # 2 * 3 = 6
    p.norm("d2")
    p.lin("x3", "e", "a")                   # 3 a
    p.free("a")
    p.norm("e")
    p.sqr("f", "e")
    p.mul("yz", "Y", "Z")
    p.free("Y", "Z")
    p.lin("dbl", "Z3", "yz", into=2)
    p.free("yz")
    p.norm("Z3")
    p.lin("subdbl", "X3", "f", "d2", into=0)  # f - 2 d : 3
    p.free("f")
    p.norm("X3")
    p.lin("x4", "c4", "c", inplace=True)    # 4 c
    p.norm("c4")
    p.lin("dbl", "c8", "c4", inplace=True)  # A = 2
    p.lin("sub", "dx", "d2", "X3", inplace=True)
    p.norm("dx")
    p.mul("t6", "e", "dx")
    p.free("e", "dx")
    p.lin("sub", "Y3", "t6", "c8", into=1)
    p.free("t6", "c8")
    p.norm("Y3")
    p.ret()
    assert sorted(p.val.items()) == [("X3", 0), ("Y3", 1), ("Z3", 2)], p.val
    return p


def g1_madd(L):
    """the generic branch of jac_madd (madd-2007-bl, gs_curve.cuh): (X, Y, Z) += (qx, qy), in place"""
    p = Prog(L, "madd")
    p.pin("X", 0), p.pin("Y", 1), p.pin("Z", 2), p.pin("qx", 3), p.pin("qy", 4)
    p.sqr("z1z1", "Z")
    p.mul("u2", "qx", "z1z1")
    p.free("qx")
    p.mul("t", "qy", "Z")
    p.free("qy")
    p.mul("s2", "t", "z1z1")
    p.free("t")
    p.lin("sub", "h", "u2", "X", inplace=True)
    p.norm("h")
    p.export_limb0("h")
    p.lin("sub", "rr0", "s2", "Y", inplace=True)
    p.lin("dbl", "rr", "rr0", inplace=True)  # 4
    p.norm("rr")
    p.sqr("hh", "h")
    p.lin("x4", "i", "hh")
    p.norm("i")
    p.mul("j", "h", "i")
    p.mul("v", "X", "i")
    p.free("X", "i")
    p.sqr("r2", "rr")
    p.lin("sub", "r2j", "r2", "j", inplace=True)
    p.lin("subdbl", "X3", "r2j", "v", into=0)  # 4
    p.free("r2j")
    p.norm("X3")
    p.lin("sub", "vx", "v", "X3", inplace=True)
    p.norm("vx")
    p.mul("t2", "rr", "vx")
    p.free("rr", "vx")
    p.mul("t3", "Y", "j")
    p.free("Y", "j")
    p.lin("subdbl", "Y3", "t2", "t3", into=1)
    p.free("t2", "t3")
    p.norm("Y3")
    p.lin("add", "zh", "Z", "h")             # A = 2
    p.free("Z", "h")
    p.sqr("t4", "zh")
    p.free("zh")
    p.lin("sub", "t5", "t4", "z1z1", inplace=True)
    p.free("z1z1")
    p.lin("sub", "Z3", "t5", "hh", into=2)
    p.free("t5", "hh")
    p.norm("Z3")
    p.ret()
    assert sorted(p.val.items()) == [("X3", 0), ("Y3", 1), ("Z3", 2)], p.val
    return p


def emit(L):
    NL = "\\n\\t"
    o = []
    progs = {"dbl": g1_dbl(L), "madd": g1_madd(L)}
    body = ["s_branch .Lgs_skipp%d_%%=" % L]
    for nm, p in progs.items():
        sym = "gs_g1_%s_sub_%d" % (nm, L)
        o.append('extern "C" __device__ void %s();' % sym)
        body += [".p2align 8", ".globl %s" % sym, ".type %s,@function" % sym, sym + ":"] + p.out
    body.append(".Lgs_skipp%d_%%=:" % L)
    o.append("// never executed: carries the subroutines' code (their labels are the symbols declared above)")
    o.append('extern "C" __device__ __attribute__((used, noinline)) void gs_g1_pointops_holder_%d() {' % L)
    o.append('  asm volatile("%s" ::: "memory");' % NL.join(body))
    o.append("}")
    mod = ['"{s%d}"(C::P28[%d])' % (40 + i, i) for i in range(L)] + ['"{s%d}"(C::P28_INV)' % (40 + L)]
    scratch_lo = 5 * L  # madd: blocks 5..13 + ACC, T are clobbered; HOUT are outputs
    # ---- doubling: X, Y, Z in / out; everything from block 3 up is clobbered
    ios = ['"+{v%d}"(%s[%d])' % (BASE + k * L + i, nm, i) for k, nm in enumerate(("x", "y", "z")) for i in range(L)]
    nout = len(ios)
    clob = ['"v%d"' % r for r in range(BASE + 3 * L, BASE + NB * L + 3)] + ['"vcc"', '"s34"', '"s35"']
    o.append("template <class C> __device__ __forceinline__ void g1_dbl_call_%d(int32_t (&x)[%d], int32_t (&y)[%d], int32_t (&z)[%d]) {"
             % (L, L, L, L))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (nout + len(mod)))
    o.append("      : %s" % ", ".join(ios))
    o.append("      : %s" % ", ".join(mod + ['"s"((uint64_t)(uintptr_t)&gs_g1_dbl_sub_%d)' % L]))
    o.append("      : %s);" % ", ".join(clob))
    o.append("}")
    # ---- mixed addition: X, Y, Z in / out, qx, qy in (destroyed), h0 out
    ios = ['"+{v%d}"(%s[%d])' % (BASE + k * L + i, nm, i) for k, nm in enumerate(("x", "y", "z", "qx", "qy")) for i in range(L)]
    ios.append('"={v%d}"(h0)' % (BASE + NB * L + 3))
    ios.append('"={v%d}"(h1)' % (BASE + NB * L + 4))
    nout = len(ios)
    clob = ['"v%d"' % r for r in range(BASE + scratch_lo, BASE + NB * L + 3)] + ['"vcc"', '"s34"', '"s35"']
    o.append("template <class C> __device__ __forceinline__ void g1_madd_call_%d(int32_t (&x)[%d], int32_t (&y)[%d], int32_t (&z)[%d], "
             "int32_t (&qx)[%d], int32_t (&qy)[%d], int32_t& h0, int32_t& h1) {" % (L, L, L, L, L, L))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (nout + len(mod)))
    o.append("      : %s" % ", ".join(ios))
    o.append("      : %s" % ", ".join(mod + ['"s"((uint64_t)(uintptr_t)&gs_g1_madd_sub_%d)' % L]))
    o.append("      : %s);" % ", ".join(clob))
    o.append("}")
    stats = {nm: (len(p.out), p.peak) for nm, p in progs.items()}
    return "\n".join(o), stats


# ======================================================================================================================
# G2 (coordinates in Fp2): the same two point operations, but their working set does not fit 256 VGPRs (a mixed
# addition has ~10 Fp2 values live = 20 blocks of L registers plus the multiplier's temporaries), so values are PARKED
# in AGPRs by an allocator that knows the whole straight-line program: farthest-next-use eviction (Belady), a value
# evicted once keeps its AGPR copy (a second eviction costs nothing), dead values are dropped.  What hipcc does for the
# same formulas is ~1 700 v_mov + ~1 000 AGPR moves + ~450 private-segment dwords per 38 multiplier calls, with two
# coordinates of the running point in the private segment across the doubling run.
#
# Register map (L = limbs), chosen so that it does not collide with the Fp2 multiplier subroutines of gs_mul28_asm.h
# (v0 .. v(7L+3)), which the caller still uses for the psi endomorphism between two calls:
#     work blocks      v0 .. v(7L-1)                   7 blocks, free inside the subroutine
#     ACC0, ACC1       v[7L : 7L+1], v[7L+2 : 7L+3]
#     running point    v(7L+4) ..  X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1      (in / out, in place)
#     addend           next 4 blocks: qx.c0 qx.c1 qy.c0 qy.c1          (madd; destroyed)
#     T, HOUT[4]       after them
#     parking          a0 .. a(10L-1)                  10 blocks
# ======================================================================================================================
class Prog2:
    NWORK, NPARK = 7, 10

    def __init__(self, L, nio):
        self.L = L
        self.nio = nio                      # I/O blocks after the accumulators (6 for dbl, 10 for madd)
        self.io_base = 7 * L + 4
        self.vblocks = [k * L for k in range(self.NWORK)] + [self.io_base + k * L for k in range(nio)]
        self.ablocks = [k * L for k in range(self.NPARK)]
        top = self.io_base + 10 * L        # same T / HOUT registers for both subroutines
        self.acc = ["v[%d:%d]" % (7 * L, 7 * L + 1), "v[%d:%d]" % (7 * L + 2, 7 * L + 3)]
        self.lo = ["v%d" % (7 * L), "v%d" % (7 * L + 2)]
        self.t = "v%d" % top
        self.hout = ["v%d" % (top + 1 + i) for i in range(4)]
        self.top = top + 5
        self.ops = []      # symbolic program
        self.inputs = {}   # value -> io block index
        self.outputs = {}  # value -> io block index
        self.out = []
        self.stats = {"park": 0, "unpark": 0, "mov": 0}

    # ---- building the symbolic program (Fq-level values; an Fp2 value is a pair of names) ---------------------------
    def inp(self, name, k):
        self.inputs[name + ".0"], self.inputs[name + ".1"] = 2 * k, 2 * k + 1
        return (name + ".0", name + ".1")

    def outp(self, val, k):
        self.outputs[val[0]], self.outputs[val[1]] = 2 * k, 2 * k + 1

    def _n(self, name):
        return (name + ".0", name + ".1")

    def mul(self, name, a, b):
        d = self._n(name)
        self.ops.append(("fp2mul", d, a, b))
        return d

    def sqr(self, name, a):
        d = self._n(name)
        self.ops.append(("fp2sqr", d, a))
        return d

    def lin(self, kind, name, a, b=None):
        d = self._n(name)
        for c in (0, 1):
            self.ops.append(("lin", kind, d[c], a[c], b[c] if b is not None else None))
        return d

    def norm(self, a):
        for c in (0, 1):
            self.ops.append(("norm", a[c]))
        return a

    def export01(self, a):
        self.ops.append(("export", a))

    # ---- allocation + emission ----------------------------------------------------------------------------------------
    def vr(self, base, i):
        return "v%d" % (base + i)

    def compile(self):
        L, ops = self.L, self.ops
        uses = {}
        for idx, op in enumerate(ops):
            for v in self._srcs(op):
                uses.setdefault(v, []).append(idx)
        final = len(ops)
        for v in self.outputs:
            uses.setdefault(v, []).append(final)  # outputs are "used" at the end
        where_v, where_a = {}, {}              # value -> VGPR block base / AGPR block base
        vfree = list(self.vblocks[:self.NWORK]) + [self.io_base + k * L for k in range(self.nio)]
        afree = list(self.ablocks)
        for v, k in self.inputs.items():
            base = self.io_base + k * L
            where_v[v] = base
            vfree.remove(base)
        self.peak_park = 0

        def next_use(v, idx):
            for u in uses.get(v, []):
                if u >= idx:
                    return u
            return None

        def drop_dead(idx):
            for v in list(where_v):
                if next_use(v, idx) is None:
                    vfree.append(where_v.pop(v))
            for v in list(where_a):
                if next_use(v, idx) is None:
                    afree.append(where_a.pop(v))

        def get_v(idx, locked, prefer=None):
            if prefer is not None and prefer in vfree:
                vfree.remove(prefer)
                return prefer
            cand = [b for b in vfree if b != prefer]
            # keep the I/O blocks of the outputs free for them where possible: temporaries take work blocks first
            cand.sort(key=lambda b: (b >= self.io_base, b))
            if cand:
                vfree.remove(cand[0])
                return cand[0]
            # evict: resident value with the farthest next use that this op does not touch
            best, far = None, -1
            for v, b in where_v.items():
                if v in locked:
                    continue
                nu = next_use(v, idx)
                nu = 10 ** 9 if nu is None else nu
                if nu > far:
                    best, far = v, nu
            assert best is not None, "no VGPR block to evict at op %d" % idx
            b = where_v.pop(best)
            if best not in where_a and next_use(best, idx) is not None:
                assert afree, "out of parking blocks at op %d" % idx
                ab = afree.pop(0)
                where_a[best] = ab
                for i in range(L):
                    self.out.append("v_accvgpr_write_b32 a%d, %s" % (ab + i, self.vr(b, i)))
                self.stats["park"] += 1
                self.peak_park = max(self.peak_park, len(where_a))
            return b

        def ensure_v(v, idx, locked):
            if v in where_v:
                return where_v[v]
            assert v in where_a, "value %s is nowhere at op %d" % (v, idx)
            b = get_v(idx, locked)
            ab = where_a[v]
            for i in range(L):
                self.out.append("v_accvgpr_read_b32 %s, a%d" % (self.vr(b, i), ab + i))
            where_v[v] = b
            self.stats["unpark"] += 1
            return b

        for idx, op in enumerate(ops):
            srcs = self._srcs(op)
            dsts = self._dsts(op)
            locked = set(srcs) | set(dsts)
            sb = {v: ensure_v(v, idx, locked) for v in srcs}
            kind = op[0]
            if kind == "fp2mul":
                _, d, a, b = op
                tmp = []
                db = []
                for c in (0, 1):
                    pref = self.io_base + self.outputs[d[c]] * L if d[c] in self.outputs else None
                    blk = get_v(idx, locked, pref)
                    where_v[d[c]] = blk
                    db.append(blk)
                n1 = get_v(idx, locked)
                tmp.append(n1)
                self._emit_fp2mul(db, [sb[a[0]], sb[a[1]]], [sb[b[0]], sb[b[1]]], n1)
                vfree.extend(tmp)
            elif kind == "fp2sqr":
                _, d, a = op
                db = []
                for c in (0, 1):
                    pref = self.io_base + self.outputs[d[c]] * L if d[c] in self.outputs else None
                    blk = get_v(idx, locked, pref)
                    where_v[d[c]] = blk
                    db.append(blk)
                tmp = [get_v(idx, locked) for _ in range(3)]
                self._emit_fp2sqr(db, [sb[a[0]], sb[a[1]]], tmp)
                vfree.extend(tmp)
            elif kind == "lin":
                _, lk, d, a, b = op
                A = sb[a]
                Bk = sb[b] if b is not None else None
                # an output goes straight to its block when that is free; otherwise in place when `a` dies here
                pref = self.io_base + self.outputs[d] * L if d in self.outputs else None
                if pref is not None and pref in vfree:
                    D = get_v(idx, locked, pref)
                elif pref is not None and where_v.get(a) == pref and next_use(a, idx + 1) is None:
                    D = where_v.pop(a)
                elif pref is None and next_use(a, idx + 1) is None and a not in where_a:
                    D = where_v.pop(a)
                else:
                    D = get_v(idx, locked, pref)
                where_v[d] = D
                self._emit_lin(lk, D, A, Bk)
            elif kind == "norm":
                self._emit_norm(sb[op[1]])
                # (the parked copy, if any, is stale now)
                if op[1] in where_a:
                    afree.append(where_a.pop(op[1]))
            elif kind == "export":
                a = op[1]
                self.out.append("v_mov_b32 %s, %s" % (self.hout[0], self.vr(sb[a[0]], 0)))
                self.out.append("v_mov_b32 %s, %s" % (self.hout[1], self.vr(sb[a[0]], 1)))
                self.out.append("v_mov_b32 %s, %s" % (self.hout[2], self.vr(sb[a[1]], 0)))
                self.out.append("v_mov_b32 %s, %s" % (self.hout[3], self.vr(sb[a[1]], 1)))
            else:
                raise ValueError(kind)
            # a lin result that overwrote a parked value's VGPR twin: nothing to do (names are SSA)
            drop_dead(idx + 1)
        # outputs into their blocks: everything else is dead by now; misplaced outputs first step aside into blocks that
        # are nobody's target (a permutation among the targets would otherwise overwrite a value not yet moved)
        drop_dead(final)
        for v in self.outputs:
            ensure_v(v, final, set(self.outputs))
        targets = {self.io_base + k * L for k in self.outputs.values()}
        misplaced = [v for v, k in self.outputs.items() if where_v[v] != self.io_base + k * L]
        for v in misplaced:
            if where_v[v] in targets:  # sits on another output's block: move aside
                spare = [b for b in vfree if b not in targets]
                assert spare, "no spare block for the final placement"
                nb = spare[0]
                vfree.remove(nb)
                for i in range(L):
                    self.out.append("v_mov_b32 %s, %s" % (self.vr(nb, i), self.vr(where_v[v], i)))
                vfree.append(where_v[v])
                where_v[v] = nb
                self.stats["mov"] += 1
        for v in misplaced:
            want = self.io_base + self.outputs[v] * L
            assert want in vfree, (v, want)
            vfree.remove(want)
            for i in range(L):
                self.out.append("v_mov_b32 %s, %s" % (self.vr(want, i), self.vr(where_v[v], i)))
            vfree.append(where_v[v])
            where_v[v] = want
            self.stats["mov"] += 1
        self.out.append("s_setpc_b64 s[34:35]")

    @staticmethod
    def _srcs(op):
        k = op[0]
        if k == "fp2mul":
            return list(dict.fromkeys(list(op[2]) + list(op[3])))
        if k == "fp2sqr":
            return list(op[2])
        if k == "lin":
            return [x for x in (op[3], op[4]) if x is not None]
        if k == "norm":
            return [op[1]]
        if k == "export":
            return list(op[1])
        raise ValueError(k)

    @staticmethod
    def _dsts(op):
        k = op[0]
        if k in ("fp2mul", "fp2sqr"):
            return list(op[1])
        if k == "lin":
            return [op[2]]
        return []

    def P(self, i):
        return "s%d" % (40 + i)

    @property
    def INV(self):
        return "s%d" % (40 + self.L)

    def _emit_fp2mul(self, d, a, b, n1):
        """c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0: two accumulators, one reduction each (gen_mul28_asm.py sub_body_fp2)"""
        L, o = self.L, self.out
        A0, A1, B0, B1 = (lambda i, x=x: self.vr(x, i) for x in (a[0], a[1], b[0], b[1]))
        R0, R1, N1 = (lambda i, x=x: self.vr(x, i) for x in (d[0], d[1], n1))
        AC0, AC1 = self.acc
        LO0, LO1 = self.lo
        assert len({a[0], a[1], d[0], d[1], n1}) == 5 and d[0] not in b and d[1] not in b and n1 not in b
        o.append(".p2align 3")
        for i in range(L):
            o.append("v_sub_u32 %s, 0, %s" % (N1(i), A1(i)))
        f0 = f1 = True
        for k in range(2 * L - 1):
            for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
                j = k - i
                o.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, A0(i), B0(j), "0" if f0 else AC0))
                f0 = False
                o.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A0(i), B1(j), "0" if f1 else AC1))
                f1 = False
                o.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, N1(i), B1(j), AC0))
                o.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A1(i), B0(j), AC1))
            for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(i), self.P(k - i), AC0))
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(i), self.P(k - i), AC1))
            if k < L:
                o.append("v_mul_lo_u32 %s, %s, %s" % (R0(k), LO0, self.INV))
                o.append("v_mul_lo_u32 %s, %s, %s" % (R1(k), LO1, self.INV))
                o.append("v_and_b32 %s, %s, %s" % (R0(k), M28, R0(k)))
                o.append("v_and_b32 %s, %s, %s" % (R1(k), M28, R1(k)))
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(k), self.P(0), AC0))
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(k), self.P(0), AC1))
            else:
                o.append("v_and_b32 %s, %s, %s" % (R0(k - L), M28, LO0))
                o.append("v_and_b32 %s, %s, %s" % (R1(k - L), M28, LO1))
            o.append("v_ashrrev_i64 %s, 28, %s" % (AC0, AC0))
            o.append("v_ashrrev_i64 %s, 28, %s" % (AC1, AC1))
        o.append("v_mov_b32 %s, %s" % (R0(L - 1), LO0))
        o.append("v_mov_b32 %s, %s" % (R1(L - 1), LO1))

    def _emit_fp2sqr(self, d, a, tmp):
        """c0 = (a0 + a1)(a0 - a1), c1 = (2 a1) a0 (gen_mul28_asm.py sub_body_fp2sqr)"""
        L, o = self.L, self.out
        A0, A1 = (lambda i, x=x: self.vr(x, i) for x in (a[0], a[1]))
        R0, R1 = (lambda i, x=x: self.vr(x, i) for x in (d[0], d[1]))
        S, D, T = (lambda i, x=x: self.vr(x, i) for x in tmp)
        AC0, AC1 = self.acc
        LO0, LO1 = self.lo
        assert len({a[0], a[1], d[0], d[1], *tmp}) == 7
        for i in range(L):
            o.append("v_add_u32 %s, %s, %s" % (S(i), A0(i), A1(i)))
            o.append("v_sub_u32 %s, %s, %s" % (D(i), A0(i), A1(i)))
            o.append("v_lshlrev_b32 %s, 1, %s" % (T(i), A1(i)))
        o.append(".p2align 3")
        f0 = f1 = True
        for k in range(2 * L - 1):
            for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
                j = k - i
                o.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, S(i), D(j), "0" if f0 else AC0))
                f0 = False
                o.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A0(i), T(j), "0" if f1 else AC1))
                f1 = False
            for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(i), self.P(k - i), AC0))
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(i), self.P(k - i), AC1))
            if k < L:
                o.append("v_mul_lo_u32 %s, %s, %s" % (R0(k), LO0, self.INV))
                o.append("v_mul_lo_u32 %s, %s, %s" % (R1(k), LO1, self.INV))
                o.append("v_and_b32 %s, %s, %s" % (R0(k), M28, R0(k)))
                o.append("v_and_b32 %s, %s, %s" % (R1(k), M28, R1(k)))
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(k), self.P(0), AC0))
                o.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(k), self.P(0), AC1))
            else:
                o.append("v_and_b32 %s, %s, %s" % (R0(k - L), M28, LO0))
                o.append("v_and_b32 %s, %s, %s" % (R1(k - L), M28, LO1))
            o.append("v_ashrrev_i64 %s, 28, %s" % (AC0, AC0))
            o.append("v_ashrrev_i64 %s, 28, %s" % (AC1, AC1))
        o.append("v_mov_b32 %s, %s" % (R0(L - 1), LO0))
        o.append("v_mov_b32 %s, %s" % (R1(L - 1), LO1))

    def _emit_lin(self, kind, D, A, Bk):
        for i in range(self.L):
            d, a = self.vr(D, i), self.vr(A, i)
            b = self.vr(Bk, i) if Bk is not None else None
            if kind == "add":
                self.out.append("v_add_u32 %s, %s, %s" % (d, a, b))
            elif kind == "sub":
                self.out.append("v_sub_u32 %s, %s, %s" % (d, a, b))
            elif kind == "dbl":
                self.out.append("v_lshlrev_b32 %s, 1, %s" % (d, a))
            elif kind == "subdbl":
                self.out.append("v_lshlrev_b32 %s, 1, %s" % (self.t, b))
                self.out.append("v_sub_u32 %s, %s, %s" % (d, a, self.t))
            elif kind == "x3":
                self.out.append("v_lshl_add_u32 %s, %s, 1, %s" % (d, a, a))
            elif kind == "x4":
                self.out.append("v_lshlrev_b32 %s, 2, %s" % (d, a))
            else:
                raise ValueError(kind)

    def _emit_norm(self, D):
        L = self.L
        for i in range(L - 1, 0, -1):
            self.out.append("v_ashrrev_i32 %s, 28, %s" % (self.t, self.vr(D, i - 1)))
            if i != L - 1:
                self.out.append("v_and_b32 %s, %s, %s" % (self.vr(D, i), M28, self.vr(D, i)))
            self.out.append("v_add_u32 %s, %s, %s" % (self.vr(D, i), self.vr(D, i), self.t))
        self.out.append("v_and_b32 %s, %s, %s" % (self.vr(D, 0), M28, self.vr(D, 0)))


def g2_dbl(L):
    p = Prog2(L, 6)
    X, Y, Z = p.inp("X", 0), p.inp("Y", 1), p.inp("Z", 2)
    a = p.sqr("a", X)
    b = p.sqr("b", Y)
    c = p.sqr("c", b)
    xb = p.lin("add", "xb", X, b)
    t2 = p.sqr("t2", xb)
    d = p.lin("sub", "d0", t2, a)
    d = p.lin("sub", "d1", d, c)
    d = p.norm(p.lin("dbl", "d", d))
    e = p.norm(p.lin("x3", "e", a))
    f = p.sqr("f", e)
    yz = p.mul("yz", Y, Z)
    Z3 = p.norm(p.lin("dbl", "Z3", yz))
    X3 = p.norm(p.lin("subdbl", "X3", f, d))
    c8 = p.lin("dbl", "c8", p.norm(p.lin("x4", "c4", c)))
    dx = p.norm(p.lin("sub", "dx", d, X3))
    t6 = p.mul("t6", e, dx)
    Y3 = p.norm(p.lin("sub", "Y3", t6, c8))
    p.outp(X3, 0), p.outp(Y3, 1), p.outp(Z3, 2)
    p.compile()
    return p


def g2_madd(L):
    p = Prog2(L, 10)
    X, Y, Z, qx, qy = p.inp("X", 0), p.inp("Y", 1), p.inp("Z", 2), p.inp("qx", 3), p.inp("qy", 4)
    z1z1 = p.sqr("z1z1", Z)
    u2 = p.mul("u2", qx, z1z1)
    t = p.mul("t", qy, Z)
    s2 = p.mul("s2", t, z1z1)
    h = p.norm(p.lin("sub", "h", u2, X))
    p.export01(h)
    rr = p.norm(p.lin("dbl", "rr", p.lin("sub", "rr0", s2, Y)))
    hh = p.sqr("hh", h)
    i = p.norm(p.lin("x4", "i", hh))
    j = p.mul("j", h, i)
    v = p.mul("v", X, i)
    r2 = p.sqr("r2", rr)
    X3 = p.norm(p.lin("subdbl", "X3", p.lin("sub", "r2j", r2, j), v))
    vx = p.norm(p.lin("sub", "vx", v, X3))
    t2 = p.mul("t2", rr, vx)
    t3 = p.mul("t3", Y, j)
    Y3 = p.norm(p.lin("subdbl", "Y3", t2, t3))
    zh = p.lin("add", "zh", Z, h)
    t4 = p.sqr("t4", zh)
    Z3 = p.norm(p.lin("sub", "Z3", p.lin("sub", "t5", t4, z1z1), hh))
    p.outp(X3, 0), p.outp(Y3, 1), p.outp(Z3, 2)
    p.compile()
    return p


def emit_g2(L):
    NL = "\\n\\t"
    o = []
    progs = {"dbl": g2_dbl(L), "madd": g2_madd(L)}
    # (the code is longer than the +-128 KB reach of s_branch: the never-executed holders return instead of jumping
    # over it, one holder per subroutine)
    for nm, p in progs.items():
        sym = "gs_g2_%s_sub_%d" % (nm, L)
        o.append('extern "C" __device__ void %s();' % sym)
        body = ["s_setpc_b64 s[30:31]", ".p2align 8", ".globl %s" % sym, ".type %s,@function" % sym, sym + ":"] + p.out
        o.append('extern "C" __device__ __attribute__((used, noinline)) void gs_g2_%s_holder_%d() {' % (nm, L))
        o.append('  asm volatile("%s" ::: "memory");' % NL.join(body))
        o.append("}")
    mod = ['"{s%d}"(C::P28[%d])' % (40 + i, i) for i in range(L)] + ['"{s%d}"(C::P28_INV)' % (40 + L)]
    pm = progs["madd"]
    io = pm.io_base
    work = ['"v%d"' % r for r in range(0, 7 * L + 4)] + ['"%s"' % pm.t]
    park = ['"a%d"' % r for r in range(Prog2.NPARK * L)]
    names6 = ("x0", "x1", "y0", "y1", "z0", "z1")
    names10 = names6 + ("qx0", "qx1", "qy0", "qy1")
    sig = lambda names: ", ".join("int32_t (&%s)[%d]" % (n, L) for n in names)
    # doubling
    ios = ['"+{v%d}"(%s[%d])' % (io + k * L + i, nm, i) for k, nm in enumerate(names6) for i in range(L)]
    o.append("template <class C> __device__ __forceinline__ void g2_dbl_call_%d(%s) {" % (L, sig(names6)))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (len(ios) + len(mod)))
    o.append("      : %s" % ", ".join(ios))
    o.append("      : %s" % ", ".join(mod + ['"s"((uint64_t)(uintptr_t)&gs_g2_dbl_sub_%d)' % L]))
    o.append("      : %s);" % ", ".join(work + park + ['"vcc"', '"s34"', '"s35"']))
    o.append("}")
    # mixed addition
    ios = ['"+{v%d}"(%s[%d])' % (io + k * L + i, nm, i) for k, nm in enumerate(names10) for i in range(L)]
    ios += ['"={%s}"(h[%d])' % (pm.hout[i], i) for i in range(4)]
    o.append("template <class C> __device__ __forceinline__ void g2_madd_call_%d(%s, int32_t (&h)[4]) {" % (L, sig(names10)))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (len(ios) + len(mod)))
    o.append("      : %s" % ", ".join(ios))
    o.append("      : %s" % ", ".join(mod + ['"s"((uint64_t)(uintptr_t)&gs_g2_madd_sub_%d)' % L]))
    o.append("      : %s);" % ", ".join(work + park + ['"vcc"', '"s34"', '"s35"']))
    o.append("}")
    stats = {nm: (len(p.out), p.stats["park"], p.stats["unpark"], p.stats["mov"]) for nm, p in progs.items()}
    return "\n".join(o), stats


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    o = ["// GENERATED by gen_pointops_asm.py -- do not edit.  (included inside namespace gs)"]
    for L in (14, 10):
        src, stats = emit(L)
        o.append("// ---- G1 point operations as subroutines, L = %d: %s" % (
            L, ", ".join("%s %d instructions (peak %d live blocks)" % (k, v[0], v[1]) for k, v in stats.items())))
        o.append(src)
        src, stats = emit_g2(L)
        o.append("// ---- G2 point operations as subroutines, L = %d: %s" % (
            L, ", ".join("%s %d instructions (%d blocks parked in AGPRs, %d fetched back, %d moved at the end)"
                         % (k, v[0], v[1], v[2], v[3]) for k, v in stats.items())))
        o.append("#if defined(GS_POINT_ASM_G2)  // measured and not shipped: see gs_curve.cuh (instruction-cache bound)")
        o.append(src)
        o.append("#endif")
    with open(os.path.join(here, "gs_pointops_asm.h"), "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote gs_pointops_asm.h")


if __name__ == "__main__":
    main()
