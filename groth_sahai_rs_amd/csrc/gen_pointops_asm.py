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
    block k = v[B+kL .. B+kL+L-1], k = 0..13, B = 48;  ACC = v[B+14L : B+14L+1], T = v[B+14L+2], HOUT = v[B+14L+3], v[B+14L+4]
    in / out:  X = block 0, Y = block 1, Z = block 2 (the running Jacobian point, updated IN PLACE)
    madd only: qx = block 3, qy = block 4 (affine addend, destroyed); HOUT = limbs 0 and 1 of the normalised
               H = U2 - X1 (the caller's cheap "H may be 0 mod p" filter: gs_fq28.cuh maybe_zero_limbs01)
    modulus in s40.. as for the multiplier subroutines, return address in s[34:35].
The edge cases of the addition (either operand the identity, H = 0) are the CALLER's: it tests before / after the call
and takes the C++ path on a saved copy (gs_curve.cuh, jac_madd_fast).
"""
import os

M28 = "0xfffffff"
NB = 14  # value blocks
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


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    o = ["// GENERATED by gen_pointops_asm.py -- do not edit.  (included inside namespace gs)"]
    for L in (14, 10):
        src, stats = emit(L)
        o.append("// ---- G1 point operations as subroutines, L = %d: %s" % (
            L, ", ".join("%s %d instructions (peak %d live blocks)" % (k, v[0], v[1]) for k, v in stats.items())))
        o.append(src)
    with open(os.path.join(here, "gs_pointops_asm.h"), "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote gs_pointops_asm.h")


if __name__ == "__main__":
    main()
