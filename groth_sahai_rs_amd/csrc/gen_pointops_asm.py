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
    madd only: qx = block 3, qy = block 4 (affine addend, destroyed); HOUT = FLAG (1: some lane may have H = U2 - X1 = 0
               mod p, nothing was touched; s[60:61] = p^-1 mod 2^56 for that test)
    modulus in s40.. as for the multiplier subroutines, return address in s[34:35].
The edge cases of the addition: an identity operand is the CALLER's (tested before the call); H = 0 is tested by the
subroutine itself, which then returns with FLAG = 1 and every operand untouched (export_limb0; gs_curve.cuh jac_madd_ip).
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
        """H = U2 - X1 = 0 mod p (P = +-Q) is the case the generic formulas do not cover: the 56-bit filter of gs_fq28.cuh
        (maybe_zero_limbs01) on the two low limbs of H, HERE -- if any active lane may have H = 0 the subroutine returns
        at once with FLAG = 1 and every operand as it came, and the caller sends the wave through the C++ addition.  (Round
        4 first handed the two limbs back and tested after the call: the caller then kept copies of both operands alive
        across every call -- seven 16-byte private-segment stores per addition step for a path taken once in 2^35.)
        s[60:61] = p^-1 mod 2^56; temporaries: the accumulator, (T, HOUT) as a pair, HOUT1, limb 0 of block NB - 1."""
        o, h = self.out, self.b(name)
        top = BASE + NB * self.L
        alo, ahi = "v%d" % top, "v%d" % (top + 1)
        assert top % 2 == 0                     # (64-bit register pairs are even-aligned on this target)
        k = "v[%d:%d]" % (top + 2, top + 3)      # (T, HOUT) as a pair
        klo, khi = self.t, self.hout
        x, y = self.r(NB - 1, 0), self.hout1
        o.append("s_mov_b32 s62, 0x10000000")
        o.append("v_mov_b32 %s, %s" % (alo, self.r(h, 0)))
        o.append("v_ashrrev_i32 %s, 31, %s" % (ahi, self.r(h, 0)))
        o.append("v_mad_i64_i32 %s, vcc, %s, s62, %s" % (self.acc, self.r(h, 1), self.acc))
        o.append("v_mul_lo_u32 %s, %s, s61" % (y, alo))
        o.append("v_mul_lo_u32 %s, %s, s60" % (x, ahi))
        o.append("v_mad_u64_u32 %s, vcc, %s, s60, 0" % (k, alo))
        o.append("v_add3_u32 %s, %s, %s, %s" % (khi, khi, y, x))
        o.append("v_bfe_i32 %s, %s, 0, 24" % (khi, khi))
        o.append("v_add_co_u32 %s, vcc, 0x100000, %s" % (klo, klo))
        o.append("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (khi, khi))
        o.append("v_cmp_eq_u32 vcc, 0, %s" % khi)
        o.append("s_mov_b64 s[38:39], vcc")
        o.append("v_cmp_ge_u32 vcc, 0x200000, %s" % klo)
        o.append("s_and_b64 vcc, vcc, s[38:39]")
        o.append("s_cbranch_vccz .Lgs_g1_go%d_%%=" % self.L)
        o.append("v_mov_b32 %s, 1" % self.hout)
        o.append("s_setpc_b64 s[34:35]")
        o.append(".Lgs_g1_go%d_%%=:" % self.L)
        self.has_flag = True

    def ret(self):
        if getattr(self, "has_flag", False):
            self.out.append("v_mov_b32 %s, 0" % self.hout)
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
    # The caller's own edge test (either operand the identity) arrives in T: if ANY active lane has it set, return at once
    # (FLAG = 1, nothing touched) -- the call itself stays UNCONDITIONAL in the caller.  (A branch around the call made
    # hipcc keep the running point in the private segment: 11 + 11 sixteen-byte accesses per addition step.)
    p.out.append("v_cmp_ne_u32 vcc, 0, %s" % p.t)
    p.out.append("s_cbranch_vccz .Lgs_g1_in%d_%%=" % L)
    p.out.append("v_mov_b32 %s, 1" % p.hout)
    p.out.append("s_setpc_b64 s[34:35]")
    p.out.append(".Lgs_g1_in%d_%%=:" % L)
    p.sqr("z1z1", "Z")
    p.mul("u2", "qx", "z1z1")
    p.mul("t", "qy", "Z")
    p.mul("s2", "t", "z1z1")
    p.free("t")
    p.lin("sub", "h", "u2", "X", inplace=True)
    p.norm("h")
    p.export_limb0("h")       # (nothing handed in has been overwritten up to here: X, Y, Z, qx, qy are all in their blocks)
    p.free("qx", "qy")
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
    ios.append('"={v%d}"(flag)' % (BASE + NB * L + 3))  # 1: an edge lane or a possible H = 0 in the wave, nothing was touched
    ios.append('"+{v%d}"(edge)' % (BASE + NB * L + 2))   # in: the caller's identity-operand test (T; destroyed)
    nout = len(ios)
    clob = ['"v%d"' % r for r in range(BASE + scratch_lo, BASE + NB * L + 2)] + ['"v%d"' % (BASE + NB * L + 4)] + \
        ['"vcc"', '"s34"', '"s35"', '"s38"', '"s39"', '"s62"']
    modf = mod + ['"{s60}"((uint32_t)(pinv56<C>() & 0xffffffffu))', '"{s61}"((uint32_t)(pinv56<C>() >> 32))']
    o.append("template <class C> __device__ __forceinline__ void g1_madd_call_%d(int32_t (&x)[%d], int32_t (&y)[%d], int32_t (&z)[%d], "
             "int32_t (&qx)[%d], int32_t (&qy)[%d], int32_t& edge, int32_t& flag) {" % (L, L, L, L, L, L))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (nout + len(modf)))
    o.append("      : %s" % ", ".join(ios))
    o.append("      : %s" % ", ".join(modf + ['"s"((uint64_t)(uintptr_t)&gs_g1_madd_sub_%d)' % L]))
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

    def inp_q(self, name, k, qk):
        """an operand of the addend (read once): the compact form takes it in AGPRs"""
        return self.inp(name, k)

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


# ======================================================================================================================
# G2, compact form (round 4, after the straight-line form lost to the instruction cache): the SAME symbolic programs, but
# every Fp2 product / squaring is a call of the SHARED subroutine of gs_mul28_asm.h (gs_fp2mul28_sub_L: a in v0.., b in
# v2L.., result in v4L.., -a1 in v6L..; gs_fp2sqr28_sub_L: a in v0.., result in v2L.., three temporaries above it), and what
# this generator owns is the data movement around them: operands copied into the subroutine's blocks (from a VGPR block or
# straight from an AGPR), results consumed where the subroutine leaves them, values that must outlive a call moved to the
# I/O blocks or parked in AGPRs by the same farthest-next-use rule.  ~1.6 k instructions of glue per addition next to the
# 13.2 k of its seven products and four squarings -- and the code is 13 KB instead of 114 KB.
#     return address   s[36:37] (the nested calls use s[34:35]);  subroutine addresses in s[56:57] (product), s[58:59]
# ======================================================================================================================
class Prog2c(Prog2):
    NPARK = 8          # parking blocks a0 .. a(8L-1); the addend's four blocks follow them

    def __init__(self, L, nio, nio_v=6):
        # nio is ignored (signature of Prog2: the program builders pass their block count); nio_v = in/out VGPR blocks.
        # the addend is read ONCE per coordinate: it arrives in AGPRs (4 blocks after the parking space) and goes straight
        # into the multiplier's operand block.  The VGPR footprint ends at v(7L+4+6L+5): the caller keeps 65 (L = 14)
        # VGPRs of its own across the call -- enough for the table entry it has requested for the NEXT step, whose loads
        # are still in flight (with the addend in VGPR I/O blocks the call clobbered all but 9 and every step waited
        # for its prefetch: k_fix.g2 12.3 -> 14.7 ms)
        Prog2.__init__(self, L, nio_v)
        top = self.io_base + nio_v * L
        self.t = "v%d" % top
        self.hout = ["v%d" % (top + 1 + i) for i in range(4)]
        self.top = top + 5
        self.ablocks = [k * L for k in range(self.NPARK)]
        self.ain_base = self.NPARK * L
        self.ainputs = {}

    def inp_q(self, name, k, qk):
        self.ainputs[name + ".0"], self.ainputs[name + ".1"] = 2 * qk, 2 * qk + 1
        return (name + ".0", name + ".1")

    SEARCH = 1500      # schedules tried per subroutine (fixed seed: the generated header is reproducible)

    def compile(self):
        """The order of the symbolic program decides how many values a call finds in the blocks it destroys: try random
        topological orders of the dependency graph (reads after their writer, an in-place carry round `norm` after every
        earlier reader of its value) and keep the one with the least data movement."""
        import random
        ops = list(self.ops)
        n = len(ops)

        def rw(op):
            k = op[0]
            if k in ("fp2mul", "fp2sqr"):
                return self._srcs(op), list(op[1])
            if k == "lin":
                return self._srcs(op), [op[2]]
            if k == "norm":
                return [op[1]], [op[1]]
            return self._srcs(op), []          # export

        pred = [set() for _ in range(n)]
        last_w, readers = {}, {}
        for i, op in enumerate(ops):
            r, w = rw(op)
            for x in r:
                if x in last_w:
                    pred[i].add(last_w[x])
            for x in w:
                if x in last_w:
                    pred[i].add(last_w[x])
                for j in readers.get(x, ()):
                    if j != i:
                        pred[i].add(j)
                last_w[x] = i
                readers[x] = []
            for x in r:
                readers.setdefault(x, []).append(i)
        rnd = random.Random(20241005 + self.L + n)
        best = None
        for trial in range(self.SEARCH + 1):
            if trial == 0:
                order = list(range(n))
            else:
                done, order = set(), []
                ready = [i for i in range(n) if not pred[i]]
                while ready:
                    # mostly follow the original order, sometimes jump: nearby schedules, not noise
                    ready.sort()
                    i = ready[0] if rnd.random() < 0.6 else rnd.choice(ready)
                    ready.remove(i)
                    order.append(i)
                    done.add(i)
                    for j in range(n):
                        if j not in done and j not in ready and pred[j] <= done:
                            ready.append(j)
                assert len(order) == n
            self.ops = [ops[i] for i in order]
            self.out, self.stats = [], {"park": 0, "unpark": 0, "mov": 0}
            try:
                self._compile_order()
            except AssertionError:
                continue
            if best is None or len(self.out) < len(best[0]):
                best = (self.out, self.stats, self.peak_park, list(self.ops))
        assert best is not None
        self.out, self.stats, self.peak_park, self.ops = best

    def _compile_order(self):
        L, ops = self.L, self.ops
        W = [k * L for k in range(7)]
        allv = list(W) + [self.io_base + k * L for k in range(self.nio)]
        uses = {}
        for idx, op in enumerate(ops):
            for v in self._srcs(op):
                uses.setdefault(v, []).append(idx)
        final = len(ops)
        for v in self.outputs:
            uses.setdefault(v, []).append(final)
        where_v, where_a = {}, {}
        afree = list(self.ablocks)
        for v, k in self.inputs.items():
            where_v[v] = self.io_base + k * L
        for v, k in self.ainputs.items():
            where_a[v] = self.ain_base + k * L    # (parking space once the coordinate has been consumed)
        self.peak_park = 0
        self.stats["calls"] = 0
        o = self.out
        o.append("s_mov_b64 s[36:37], s[34:35]")
        # Until the "export" op (the H = 0 test of the addition) NOTHING the caller handed in may be overwritten: the
        # subroutine returns early from there with every operand intact (see the op).  -1: no such op (doubling).
        exp_idx = max([i for i, op in enumerate(ops) if op[0] == "export"] + [-1])
        io_blocks = {self.io_base + k * L for k in range(self.nio)}
        deferred_a = []

        def next_use(v, idx):
            for u in uses.get(v, []):
                if u >= idx:
                    return u
            return None

        def free_blocks():
            occ = set(where_v.values())
            return [b for b in allv if b not in occ]

        def drop_dead(idx):
            for v in list(where_v):
                if next_use(v, idx) is None:
                    where_v.pop(v)
            for v in list(where_a):
                if next_use(v, idx) is None:
                    ab = where_a.pop(v)            # (the addend's blocks too: they are in/out operands, "destroyed")
                    if ab >= self.ain_base and idx <= exp_idx:
                        deferred_a.append(ab)      # ... but only after the early-exit point
                    else:
                        afree.append(ab)
            if idx > exp_idx and deferred_a:
                afree.extend(deferred_a)
                del deferred_a[:]

        def park(v):
            """VGPR copy of v goes away; make sure an AGPR copy exists"""
            b = where_v.pop(v)
            if v not in where_a:
                assert afree, "out of parking blocks"
                ab = afree.pop(0)
                where_a[v] = ab
                for i in range(L):
                    o.append("v_accvgpr_write_b32 a%d, %s" % (ab + i, self.vr(b, i)))
                self.stats["park"] += 1
                self.peak_park = max(self.peak_park, len(where_a))
            return b

        def get_v(idx, locked, avoid=(), prefer=None):
            if idx <= exp_idx:
                avoid = set(avoid) | io_blocks
            fr = [b for b in free_blocks() if b not in avoid]
            if prefer is not None and prefer in fr:
                return prefer
            fr.sort(key=lambda b: (b < self.io_base, b))      # I/O blocks first: the work blocks turn over at every call
            if fr:
                return fr[0]
            best, far = None, -1
            for v, b in where_v.items():
                if v in locked or b in avoid:
                    continue
                nu = next_use(v, idx)
                nu = 10 ** 9 if nu is None else nu
                if nu > far:
                    best, far = v, nu
            assert best is not None, "no VGPR block to evict at op %d" % idx
            if next_use(best, idx) is None:
                return where_v.pop(best)
            return park(best)

        def ensure_v(v, idx, locked, avoid=()):
            if v in where_v:
                return where_v[v]
            assert v in where_a, "value %s is nowhere at op %d" % (v, idx)
            b = get_v(idx, locked, avoid)
            ab = where_a[v]
            for i in range(L):
                o.append("v_accvgpr_read_b32 %s, a%d" % (self.vr(b, i), ab + i))
            where_v[v] = b
            self.stats["unpark"] += 1
            return b

        def mov(dst, src):
            for i in range(L):
                o.append("v_mov_b32 %s, %s" % (self.vr(dst, i), self.vr(src, i)))
            self.stats["mov"] += 1

        def call(idx, operands, nclob, results, addr):
            """operands: values wanted in W[0..]; the call destroys W[len(operands) .. len(operands)+nclob-1] (results first)"""
            n = len(operands)
            tgt = {W[k]: operands[k] for k in range(n)}
            blocks = set(W[:n + nclob])
            locked = set(operands)
            # 1. values the call would destroy and that are needed later (or are no operand of it) leave its blocks
            for v, b in list(where_v.items()):
                if b not in blocks:
                    continue
                if v in operands:
                    continue                      # the parallel move below takes care of it
                if next_use(v, idx + 1) is None and next_use(v, idx) is None:
                    where_v.pop(v)
                    continue
                fr = [x for x in free_blocks() if x not in blocks]
                if fr and v not in where_a:
                    fr.sort()
                    mov(fr[0], b)
                    where_v[v] = fr[0]
                else:
                    park(v)
            # an operand that must survive the call but sits in a block the call destroys, and is not moved into a
            # preserved block by the call's own placement: same treatment (it then comes back from where it went)
            # 2. parallel move of the operands into W[0..n-1]
            pending = {}
            for t, v in tgt.items():
                if where_v.get(v) == t:
                    continue
                pending[t] = v
            # operands living in a destroyed block that are needed after the call and are NOT about to be moved to an
            # operand block: cannot happen (every operand is moved into or already sits in an operand block)
            guard = 0
            while pending:
                guard += 1
                assert guard < 64
                srcblocks = {where_v[v] for v in pending.values() if v in where_v}
                done = False
                for t, v in list(pending.items()):
                    if t in srcblocks and where_v.get(v) != t:
                        # t still holds a value somebody else wants moved out first
                        holder = [u for u in pending.values() if where_v.get(u) == t]
                        if holder:
                            continue
                    if v in where_v:
                        s_ = where_v[v]
                        mov(t, s_)
                        if s_ in blocks or next_use(v, idx + 1) is None:
                            where_v[v] = t          # moved (or dead after the call anyway)
                        # else: copied; the original stays the value's home, the copy is transient
                        elif True:
                            pass
                    else:
                        ab = where_a[v]
                        for i in range(L):
                            o.append("v_accvgpr_read_b32 %s, a%d" % (self.vr(t, i), ab + i))
                        self.stats["unpark"] += 1
                        if next_use(v, idx + 1) is None:
                            pass
                    pending.pop(t)
                    done = True
                    break
                if not done:
                    # a cycle among the operand blocks: one of them steps aside into the last block the call destroys
                    t, v = next(iter(pending.items()))
                    u = [x for x in pending.values() if where_v.get(x) == t][0]
                    tmp = W[n + nclob - 1]
                    assert tmp not in where_v.values()
                    mov(tmp, t)
                    where_v[u] = tmp
            # transient copies must not be mistaken for homes: a value whose home is outside `blocks` keeps it
            o.append("s_swappc_b64 s[34:35], %s" % addr)
            self.stats["calls"] += 1
            # whatever still claims a destroyed block is gone now (operands with no later use)
            for v, b in list(where_v.items()):
                if b in set(W[n:n + nclob]):
                    assert next_use(v, idx + 1) is None, (v, idx)
                    where_v.pop(v)
            for k, d in enumerate(results):
                where_v[d] = W[n + k]

        for idx, op in enumerate(ops):
            kind = op[0]
            if kind == "fp2mul":
                _, d, a, b = op
                call(idx, [a[0], a[1], b[0], b[1]], 3, [d[0], d[1]], "s[56:57]")
            elif kind == "fp2sqr":
                _, d, a = op
                call(idx, [a[0], a[1]], 5, [d[0], d[1]], "s[58:59]")
            elif kind == "lin":
                _, lk, d, a, b = op
                locked = {x for x in (a, b, d) if x is not None}
                A = ensure_v(a, idx, locked)
                Bk = ensure_v(b, idx, locked) if b is not None else None
                pref = self.io_base + self.outputs[d] * L if d in self.outputs else None
                fr = free_blocks()
                if pref is not None and pref in fr:
                    D = pref
                elif next_use(a, idx + 1) is None and a not in where_a and (pref is None or where_v.get(a) == pref) \
                        and not (idx <= exp_idx and where_v.get(a) in io_blocks):
                    D = where_v.pop(a)
                else:
                    D = get_v(idx, locked, prefer=pref)
                if idx <= exp_idx:
                    assert D not in io_blocks
                where_v[d] = D
                self._emit_lin(lk, D, A, Bk)
            elif kind == "norm":
                blk = ensure_v(op[1], idx, {op[1]})
                if idx <= exp_idx:
                    assert blk not in io_blocks
                self._emit_norm(blk)
                if op[1] in where_a:
                    afree.append(where_a.pop(op[1]))
            elif kind == "export":
                # H = U2 - X1 = 0 mod p (P = +-Q) is the case the generic formulas do not cover.  The 56-bit filter of
                # gs_fq28.cuh (maybe_zero_limbs01: k = (V mod 2^56) p^-1 mod 2^56 is a SMALL signed integer when V = k p)
                # on both coordinates of H, here: if ANY active lane may have H = 0, the subroutine returns at once with
                # FLAG = 1 and every operand as it came (point in its I/O blocks, addend in its AGPR blocks), and the caller
                # sends those lanes through the C++ addition.  False alarms 2^-35 per lane.  Nothing is live in the caller
                # across the call but the two operands -- with the test outside, hipcc kept copies of both (140 registers)
                # and spilled the table entry it had just requested for the next step.
                # s[60:61] = p^-1 mod 2^56 (low 32, high 24 bits); temporaries: the accumulators, T, HOUT[0..3]
                a = op[1]
                b0, b1 = ensure_v(a[0], idx, set(a)), ensure_v(a[1], idx, set(a))
                a0lo, a0hi = "v%d" % (7 * L), "v%d" % (7 * L + 1)
                a1lo, a1hi = "v%d" % (7 * L + 2), "v%d" % (7 * L + 3)
                T, H1 = self.t, self.hout[1]
                o.append("s_mov_b32 s62, 0x10000000")
                for c, blk in enumerate((b0, b1)):
                    o.append("v_mov_b32 %s, %s" % (a0lo, self.vr(blk, 0)))
                    o.append("v_ashrrev_i32 %s, 31, %s" % (a0hi, self.vr(blk, 0)))
                    o.append("v_mad_i64_i32 %s, vcc, %s, s62, %s" % (self.acc[0], self.vr(blk, 1), self.acc[0]))
                    o.append("v_mul_lo_u32 %s, %s, s61" % (T, a0lo))
                    o.append("v_mul_lo_u32 %s, %s, s60" % (H1, a0hi))
                    o.append("v_mad_u64_u32 %s, vcc, %s, s60, 0" % (self.acc[1], a0lo))
                    o.append("v_add3_u32 %s, %s, %s, %s" % (a1hi, a1hi, T, H1))
                    o.append("v_bfe_i32 %s, %s, 0, 24" % (a1hi, a1hi))
                    o.append("v_add_co_u32 %s, vcc, 0x100000, %s" % (a1lo, a1lo))
                    o.append("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (a1hi, a1hi))
                    o.append("v_cmp_eq_u32 vcc, 0, %s" % a1hi)
                    o.append("s_mov_b64 s[38:39], vcc" if c == 0 else "s_and_b64 s[38:39], s[38:39], vcc")
                    o.append("v_cmp_ge_u32 vcc, 0x200000, %s" % a1lo)
                    o.append("s_and_b64 s[38:39], s[38:39], vcc")
                o.append("s_mov_b64 vcc, s[38:39]")
                o.append("s_cbranch_vccz .Lgs_g2c_go%d_%%=" % L)
                o.append("v_mov_b32 %s, 1" % self.hout[0])
                o.append("s_setpc_b64 s[36:37]")
                o.append(".Lgs_g2c_go%d_%%=:" % L)
            else:
                raise ValueError(kind)
            drop_dead(idx + 1)
        # outputs into their blocks (everything else is dead): misplaced ones sitting on another output's block step aside
        drop_dead(final)
        for v in self.outputs:
            ensure_v(v, final, set(self.outputs))
        targets = {self.io_base + k * L for k in self.outputs.values()}
        misplaced = [v for v, k in self.outputs.items() if where_v[v] != self.io_base + k * L]
        for v in misplaced:
            if where_v[v] in targets:
                spare = [b for b in free_blocks() if b not in targets]
                assert spare, "no spare block for the final placement"
                mov(spare[0], where_v[v])
                where_v[v] = spare[0]
        for v in misplaced:
            want = self.io_base + self.outputs[v] * L
            assert want in free_blocks(), (v, want)
            mov(want, where_v[v])
            where_v[v] = want
        if exp_idx >= 0:
            o.append("v_mov_b32 %s, 0" % self.hout[0])
        o.append("s_setpc_b64 s[36:37]")


_G2_CACHE = {}


def _cached(fn):
    def wrapper(L, cls=None):
        key = (fn.__name__, L, cls)
        if key not in _G2_CACHE:
            _G2_CACHE[key] = fn(L, cls)
        return _G2_CACHE[key]
    wrapper.__name__ = fn.__name__
    return wrapper


@_cached
def g2_dbl(L, cls=None):
    p = (cls or Prog2)(L, 6)
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


@_cached
def g2_madd(L, cls=None):
    p = (cls or Prog2)(L, 10)
    X, Y, Z, qx, qy = p.inp("X", 0), p.inp("Y", 1), p.inp("Z", 2), p.inp_q("qx", 3, 0), p.inp_q("qy", 4, 1)
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


# ---- a tower program on the same machinery (round 4): the general Fp6 product ------------------------------------------
# (L = 14 is BLS12-381: xi = 1 + u;  L = 10 is BN254: xi = 9 + u -- the wrappers static_assert it)
def _mul_xi(p, name, a, L, normed_in):
    """xi * a as Fq-level linear ops (gs_tower.cuh mul_xi): returns the new value (lazy)"""
    d = p._n(name)
    if L == 14:   # (1 + u)(a0 + a1 u) = (a0 - a1) + (a0 + a1) u
        p.ops.append(("lin", "sub", d[0], a[0], a[1]))
        p.ops.append(("lin", "add", d[1], a[0], a[1]))
        return d
    # (9 + u)(a0 + a1 u) = (9 a0 - a1) + (9 a1 + a0) u ; 8x is normalised on the way (input A <= 1.9)
    e = p._n(name + "e")
    for c in (0, 1):
        p.ops.append(("lin", "x4", e[c], a[c], None))
        p.ops.append(("norm", e[c]))
        p.ops.append(("lin", "dbl", e[c] + "d", e[c], None))
        p.ops.append(("norm", e[c] + "d"))
    e8 = (e[0] + "d", e[1] + "d")
    s0, s1 = name + "s.0", name + "s.1"
    p.ops.append(("lin", "add", s0, e8[0], a[0]))
    p.ops.append(("lin", "sub", d[0], s0, a[1]))
    p.ops.append(("lin", "add", s1, e8[1], a[1]))
    p.ops.append(("lin", "add", d[1], s1, a[0]))
    return d


@_cached
def f6_mul(L, cls=None):
    """gs_tower.cuh f6_mul, statement by statement: a (in/out VGPR blocks) *= b (AGPR blocks)"""
    p = Prog2c(L, 6, 6)
    a = [p.inp("a%d" % k, k) for k in range(3)]
    b = [p.inp_q("b%d" % k, 0, k) for k in range(3)]
    v0, v1, v2 = p.mul("v0", a[0], b[0]), p.mul("v1", a[1], b[1]), p.mul("v2", a[2], b[2])

    def cross(nm, i, j, vi, vj):
        sa, sb = p.lin("add", nm + "sa", a[i], a[j]), p.lin("add", nm + "sb", b[i], b[j])
        m = p.mul(nm + "m", sa, sb)
        return p.lin("sub", nm, p.lin("sub", nm + "x", m, vi), vj)
    t0 = cross("t0", 1, 2, v1, v2)
    t1 = cross("t1", 0, 1, v0, v1)
    t2 = cross("t2", 0, 2, v0, v2)
    if L != 14:
        t0 = p.norm(t0)
    r0 = p.norm(p.lin("add", "r0", v0, _mul_xi(p, "xt0", t0, L, L != 14)))
    r1 = p.norm(p.lin("add", "r1", t1, _mul_xi(p, "xv2", v2, L, True)))
    r2 = p.norm(p.lin("add", "r2", t2, v1))
    p.outp(r0, 0), p.outp(r1, 1), p.outp(r2, 2)
    p.compile()
    return p


# (Karabina's compressed squaring was generated too and dropped: its 22 carry rounds and 30 linear operations are ~1 640
# instructions whichever compiler arranges them; with the ~40 block moves around its four calls the generated form came to
# 2 180 instructions against the ~1 950 hipcc's loop body spends between the same calls, with no memory access in either.)


def emit_tower(L):
    """C++ wrapper of the tower subroutine (compact form only)"""
    NL = "\\n\\t"
    o = []
    mod = ['"{s%d}"(C::P28[%d])' % (40 + i, i) for i in range(L)] + ['"{s%d}"(C::P28_INV)' % (40 + L)]
    mod += ['"{s[56:57]}"((uint64_t)(uintptr_t)&gs_fp2mul28_sub_%d)' % L,
            '"{s[58:59]}"((uint64_t)(uintptr_t)&gs_fp2sqr28_sub_%d)' % L]
    stats = {}
    for nm, prog, vnames, anames in (
            ("f6mul", f6_mul(L), ["a00", "a01", "a10", "a11", "a20", "a21"], ["b00", "b01", "b10", "b11", "b20", "b21"]),):
        sym = "gs_%s_sub_%d" % (nm, L)
        o.append('extern "C" __device__ void %s();' % sym)
        body = ["s_setpc_b64 s[30:31]", ".p2align 8", ".globl %s" % sym, ".type %s,@function" % sym, sym + ":"] + prog.out
        o.append('extern "C" __device__ __attribute__((used, noinline)) void gs_%s_holder_%d() {' % (nm, L))
        o.append('  asm volatile("%s" ::: "memory");' % NL.join(body))
        o.append("}")
        io = prog.io_base
        ios = ['"+{v%d}"(%s[%d])' % (io + k * L + i, n_, i) for k, n_ in enumerate(vnames) for i in range(L)]
        ios += ['"+{a%d}"(%s[%d])' % (prog.ain_base + k * L + i, n_, i) for k, n_ in enumerate(anames) for i in range(L)]
        work = ['"v%d"' % r for r in range(0, 7 * L + 4)] + ['"%s"' % prog.t] + ['"s36"', '"s37"']
        park = ['"a%d"' % r for r in range(Prog2c.NPARK * L)]
        sig = ", ".join("int32_t (&%s)[%d]" % (n_, L) for n_ in vnames + anames)
        o.append("template <class C> __device__ __forceinline__ void %s_call_%d(%s) {" % (nm, L, sig))
        o.append("  static_assert(C::XI_A == %d, \"the generated tower code is per (limb count, xi)\");" % (1 if L == 14 else 9))
        o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (len(ios) + len(mod)))
        o.append("      : %s" % ", ".join(ios))
        o.append("      : %s" % ", ".join(mod + ['"s"((uint64_t)(uintptr_t)&%s)' % sym]))
        o.append("      : %s);" % ", ".join(work + park + ['"vcc"', '"s34"', '"s35"']))
        o.append("}")
        stats[nm] = (len(prog.out), prog.stats["park"], prog.stats["unpark"], prog.stats["mov"], prog.stats["calls"])
    return "\n".join(o), stats


def emit_g2(L, cls=None):
    NL = "\\n\\t"
    o = []
    compact = cls is Prog2c
    progs = {"dbl": g2_dbl(L, cls), "madd": g2_madd(L, cls)}
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
    if compact:  # the shared multiplier subroutines' addresses, where the generated code expects them
        mod += ['"{s[56:57]}"((uint64_t)(uintptr_t)&gs_fp2mul28_sub_%d)' % L,
                '"{s[58:59]}"((uint64_t)(uintptr_t)&gs_fp2sqr28_sub_%d)' % L]
    pm = progs["madd"]
    io = pm.io_base
    work = ['"v%d"' % r for r in range(0, 7 * L + 4)] + ['"%s"' % pm.t]
    if compact:
        work += ['"s36"', '"s37"', '"s38"', '"s39"', '"s62"'] + ['"%s"' % h for h in pm.hout[1:]]
        mod += ['"{s60}"((uint32_t)(pinv56<C>() & 0xffffffffu))', '"{s61}"((uint32_t)(pinv56<C>() >> 32))']
    park = ['"a%d"' % r for r in range((Prog2c if compact else Prog2).NPARK * L)]
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
    if compact:  # the addend arrives in the four AGPR blocks after the parking space (and is destroyed)
        ios = ['"+{v%d}"(%s[%d])' % (io + k * L + i, nm, i) for k, nm in enumerate(names6) for i in range(L)]
        ios += ['"+{a%d}"(%s[%d])' % (pm.ain_base + k * L + i, nm, i) for k, nm in enumerate(names10[6:]) for i in range(L)]
    else:
        ios = ['"+{v%d}"(%s[%d])' % (io + k * L + i, nm, i) for k, nm in enumerate(names10) for i in range(L)]
    if compact:  # FLAG: 1 = "some lane may have H = 0: nothing was touched, take the C++ addition"
        ios += ['"={%s}"(flag)' % pm.hout[0]]
        o.append("template <class C> __device__ __forceinline__ void g2_madd_call_%d(%s, int32_t& flag) {" % (L, sig(names10)))
    else:
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
        o.append("#if defined(GS_POINT_ASM_G2_STRAIGHT)  // measured and not shipped: see gs_curve.cuh (instruction-cache bound)")
        o.append(src)
        src, stats = emit_g2(L, Prog2c)
        o.append("#elif defined(GS_POINT_ASM_G2)")
        o.append("// ---- G2 point operations, compact form (calls of the shared Fp2 subroutines), L = %d: %s" % (
            L, ", ".join("%s %d instructions of glue (%d parked, %d fetched back, %d block moves)"
                         % (k, v[0], v[1], v[2], v[3]) for k, v in stats.items())))
        o.append(src)
        o.append("#endif")
        if not os.environ.get("GS_TOWER_ASM_EMIT"):   # measured and not shipped (gs_tower.cuh: f12_mul)
            continue
        src, stats = emit_tower(L)
        o.append("// ---- tower operations as subroutines (compact form), L = %d: %s" % (
            L, ", ".join("%s %d instructions of glue around %d calls (%d parked, %d fetched back, %d block moves)"
                         % (k, v[0], v[4], v[1], v[2], v[3]) for k, v in stats.items())))
        o.append("#define GS_TOWER_ASM 1")
        o.append(src)
    with open(os.path.join(here, "gs_pointops_asm.h"), "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote gs_pointops_asm.h")


if __name__ == "__main__":
    main()
