#!/usr/bin/env python3
"""Generate gs_mul28_asm.h: radix-2^28 Montgomery product for gfx950 as one inline
asm block per limb count (L = 14: BLS12-381 Fq; L = 10: BN254 Fq).

Product scanning with ONE signed 64-bit accumulator v[32:33]:
    column k:  acc += sum a_i b_(k-i)   (v_mad_i64_i32)
               acc += sum m_i p_(k-i)   (v_mad_u64_u32, i < k)
    k <  L:    m_k = (acc * (-p^-1)) & M ; acc += m_k p_0 ; acc >>= 28
    k >= L:    r_(k-L) = acc & M ; acc >>= 28          (last: r_(L-1) = acc)
No carry-flag instruction anywhere (they are half rate on gfx950); L*L*2 mads
+ ~5 full-rate ops per column.  Every instruction of the block is 8 bytes long and the
block starts 8-byte aligned (.p2align 3): with the stream at 4 (mod 8) the SAME code ran
10 % slower end to end (measured; the placement used to flip with unrelated edits).  m_k lives in the register that later receives
r_k.  v32..v34 are caller-saved scratch VGPRs (the product is out of line).
Included INSIDE namespace gs by gs_fq28.cuh.
"""
import os


def gen(L):
    R = lambda i: "%%%d" % i
    A = lambda i: "%%%d" % (L + i)
    B = lambda i: "%%%d" % (2 * L + i)
    P = lambda i: "%%%d" % (3 * L + i)
    INV = "%%%d" % (4 * L)
    ACC, LO, T = "v[32:33]", "v32", "v34"
    out = []
    first = True
    for k in range(2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            src2 = "0" if first else ACC
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC, A(i), B(k - i), src2))
            first = False
        for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, R(i), P(k - i), ACC))
        if k < L:
            # m_k = (acc * (-p^-1)) mod 2^28: the low 28 bits of a product depend only on the operands' low 28 bits,
            # so the accumulator's low word goes in unmasked
            out.append("v_mul_lo_u32 %s, %s, %s" % (R(k), LO, INV))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R(k), R(k)))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, R(k), P(0), ACC))
        else:
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R(k - L), LO))
        out.append("v_ashrrev_i64 %s, 28, %s" % (ACC, ACC))
    out.append("v_mov_b32 %s, %s" % (R(L - 1), LO))
    body = ".p2align 3\\n\\t" + "\\n\\t".join(out)
    outs = ", ".join('"=&v"(r[%d])' % i for i in range(L))
    ins = ", ".join('"v"(a[%d])' % i for i in range(L)) + ", " + ", ".join('"v"(b[%d])' % i for i in range(L))
    ins += ", " + ", ".join('"s"(C::P28[%d])' % i for i in range(L)) + ', "s"(C::P28_INV)'
    clob = '"vcc", "v32", "v33", "v34"'
    return (
        "template <class C> __device__ __forceinline__ void mul28_asm_%d(int32_t (&r)[%d], const int32_t (&a)[%d], "
        "const int32_t (&b)[%d]) {\n  asm(\"%s\"\n      : %s\n      : %s\n      : %s);\n}\n" % (L, L, L, L, body, outs, ins, clob),
        len(out),
    )


def gen_sqr(L):
    """a^2: the cross products a_i a_j (i < j) are taken once against the doubled operand d = 2a, the diagonal ones
    against a itself: L(L+1)/2 product mads instead of L^2.  Operands: r (L outputs), a (L), d (L, = 2a), p, inv."""
    R = lambda i: "%%%d" % i
    A = lambda i: "%%%d" % (L + i)
    D = lambda i: "%%%d" % (2 * L + i)
    P = lambda i: "%%%d" % (3 * L + i)
    INV = "%%%d" % (4 * L)
    ACC, LO, T = "v[32:33]", "v32", "v34"
    out = []
    first = True
    for k in range(2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            j = k - i
            if i > j:
                continue
            src2 = "0" if first else ACC
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC, A(i), A(j) if i == j else D(j), src2))
            first = False
        for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, R(i), P(k - i), ACC))
        if k < L:
            # m_k = (acc * (-p^-1)) mod 2^28: the low 28 bits of a product depend only on the operands' low 28 bits,
            # so the accumulator's low word goes in unmasked
            out.append("v_mul_lo_u32 %s, %s, %s" % (R(k), LO, INV))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R(k), R(k)))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, R(k), P(0), ACC))
        else:
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R(k - L), LO))
        out.append("v_ashrrev_i64 %s, 28, %s" % (ACC, ACC))
    out.append("v_mov_b32 %s, %s" % (R(L - 1), LO))
    body = ".p2align 3\\n\\t" + "\\n\\t".join(out)
    outs = ", ".join('"=&v"(r[%d])' % i for i in range(L))
    ins = ", ".join('"v"(a[%d])' % i for i in range(L)) + ", " + ", ".join('"v"(d[%d])' % i for i in range(L))
    ins += ", " + ", ".join('"s"(C::P28[%d])' % i for i in range(L)) + ', "s"(C::P28_INV)'
    clob = '"vcc", "v32", "v33", "v34"'
    return (
        "template <class C> __device__ __forceinline__ void sqr28_asm_%d(int32_t (&r)[%d], const int32_t (&a)[%d], "
        "const int32_t (&d)[%d]) {\n  asm(\"%s\"\n      : %s\n      : %s\n      : %s);\n}\n" % (L, L, L, L, body, outs, ins, clob),
        len(out),
    )


# ---- the same bodies as SUBROUTINES reached with s_swappc_b64 from an inline-asm "call" --------------------------------
# A C++ function call costs more than its branch: every non-kernel function starts with s_waitcnt vmcnt(0) lgkmcnt(0)
# (the callee cannot know what is in flight), so each of the ~10^5 multiplier calls of a pairing lane also waited for
# every spill store and operand load its caller had just issued.  Calling the multiplier from an asm statement with
# FIXED operand registers keeps the hardware call (s_swappc_b64 / s_setpc_b64) and drops the wait: the compiler sees
# an instruction that reads v0..v(2L-1) and the modulus in s40.., writes v(2L)..v(3L-1) and clobbers three VGPRs, vcc
# and s[34:35], and inserts only the waits those registers need.  The subroutine bodies live behind labels inside a
# never-executed holder function (file-scope asm is not emitted for the device).
def sub_body(L, square):
    A = lambda i: "v%d" % i
    B = lambda i: "v%d" % (L + i)
    R = lambda i: "v%d" % (2 * L + i)
    P = lambda i: "s%d" % (40 + i)
    INV = "s%d" % (40 + L)
    ACC, LO = "v[%d:%d]" % (3 * L, 3 * L + 1), "v%d" % (3 * L)
    out = []
    if square:  # d = 2a in the (clobbered) b registers
        for i in range(L):
            out.append("v_lshlrev_b32 %s, 1, %s" % (B(i), A(i)))
    first = True
    for k in range(2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            j = k - i
            if square and i > j:
                continue
            y = (A(j) if i == j else B(j)) if square else B(j)
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC, A(i), y, "0" if first else ACC))
            first = False
        for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, R(i), P(k - i), ACC))
        if k < L:
            out.append("v_mul_lo_u32 %s, %s, %s" % (R(k), LO, INV))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R(k), R(k)))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC, R(k), P(0), ACC))
        else:
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R(k - L), LO))
        out.append("v_ashrrev_i64 %s, 28, %s" % (ACC, ACC))
    out.append("v_mov_b32 %s, %s" % (R(L - 1), LO))
    out.append("s_setpc_b64 s[34:35]")
    return out


# ---- Fp2 product as ONE subroutine: schoolbook with lazy reduction ------------------------------------------------------
#   c0 = a0 b0 - a1 b1,  c1 = a0 b1 + a1 b0
# Each output is TWO product-scanning sums in ONE accumulator (a1 enters c0 negated) followed by ONE Montgomery
# reduction: 4 L^2 + 2 L^2 mads = the mads of Karatsuba's three full products, but none of Karatsuba's 5 lazy
# additions, no carry round on the result, a third less marshalling, and both outputs come out normalised.
# Accumulator bound: 2 L products + L reduction terms per column, |a_i b_j| <= A_a A_b 2^56, so 28 A_a A_b + 14 < 128:
# A_a A_b <= 4 -- a product of two sums of two normalised values (what mul_l2 is for) fits, with ~1.5 % to spare.
# Registers: a0 v0.., a1 vL.., b0 v2L.., b1 v3L.. (preserved) -> c0 v4L.., c1 v5L..; -a1 v6L.., two accumulators and
# nothing else; modulus in s40.. as for the single product.
def sub_body_fp2(L):
    A0 = lambda i: "v%d" % i
    A1 = lambda i: "v%d" % (L + i)
    B0 = lambda i: "v%d" % (2 * L + i)
    B1 = lambda i: "v%d" % (3 * L + i)
    R0 = lambda i: "v%d" % (4 * L + i)
    R1 = lambda i: "v%d" % (5 * L + i)
    N1 = lambda i: "v%d" % (6 * L + i)
    P = lambda i: "s%d" % (40 + i)
    INV = "s%d" % (40 + L)
    AC0, LO0 = "v[%d:%d]" % (7 * L, 7 * L + 1), "v%d" % (7 * L)
    AC1, LO1 = "v[%d:%d]" % (7 * L + 2, 7 * L + 3), "v%d" % (7 * L + 2)
    out = []
    for i in range(L):
        out.append("v_sub_u32 %s, 0, %s" % (N1(i), A1(i)))
    f0 = f1 = True
    for k in range(2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            j = k - i
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, A0(i), B0(j), "0" if f0 else AC0))
            f0 = False
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A0(i), B1(j), "0" if f1 else AC1))
            f1 = False
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, N1(i), B1(j), AC0))
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A1(i), B0(j), AC1))
        for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(i), P(k - i), AC0))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(i), P(k - i), AC1))
        if k < L:
            out.append("v_mul_lo_u32 %s, %s, %s" % (R0(k), LO0, INV))
            out.append("v_mul_lo_u32 %s, %s, %s" % (R1(k), LO1, INV))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R0(k), R0(k)))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R1(k), R1(k)))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(k), P(0), AC0))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(k), P(0), AC1))
        else:
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R0(k - L), LO0))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R1(k - L), LO1))
        out.append("v_ashrrev_i64 %s, 28, %s" % (AC0, AC0))
        out.append("v_ashrrev_i64 %s, 28, %s" % (AC1, AC1))
    out.append("v_mov_b32 %s, %s" % (R0(L - 1), LO0))
    out.append("v_mov_b32 %s, %s" % (R1(L - 1), LO1))
    out.append("s_setpc_b64 s[34:35]")
    return out


def gen_fp2_call(L):
    NLT = "\\n\\t"
    sym = "gs_fp2mul28_sub_%d" % L
    o = ['extern "C" __device__ void %s();' % sym]
    body = ["s_branch .Lgs_skipf%d_%%=" % L, ".p2align 8", ".globl %s" % sym, ".type %s,@function" % sym, sym + ":"]
    body += sub_body_fp2(L) + [".Lgs_skipf%d_%%=:" % L]
    o.append('extern "C" __device__ __attribute__((used, noinline)) void gs_fp2mul28_sub_holder_%d() {' % L)
    o.append('  asm volatile("%s" ::: "memory");' % NLT.join(body))
    o.append("}")
    outs = ", ".join('"={v%d}"(r0[%d])' % (4 * L + i, i) for i in range(L)) + ", " + \
        ", ".join('"={v%d}"(r1[%d])' % (5 * L + i, i) for i in range(L))
    ins = []
    for base, nm in ((0, "a0"), (L, "a1"), (2 * L, "b0"), (3 * L, "b1")):
        ins += ['"{v%d}"(%s[%d])' % (base + i, nm, i) for i in range(L)]
    ins += ['"{s%d}"(C::P28[%d])' % (40 + i, i) for i in range(L)] + ['"{s%d}"(C::P28_INV)' % (40 + L)]
    nin = 4 * L + L + 1
    ins.append('"s"((uint64_t)(uintptr_t)&%s)' % sym)
    clob = ['"v%d"' % (6 * L + i) for i in range(L + 4)] + ['"vcc"', '"s34"', '"s35"']
    o.append("template <class C> __device__ __forceinline__ void fp2mul28_call_%d(int32_t (&r0)[%d], int32_t (&r1)[%d], "
             "const int32_t (&a0)[%d], const int32_t (&a1)[%d], const int32_t (&b0)[%d], const int32_t (&b1)[%d]) {"
             % (L, L, L, L, L, L, L))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (2 * L + nin))
    o.append("      : %s" % outs)
    o.append("      : %s" % ", ".join(ins))
    o.append("      : %s);" % ", ".join(clob))
    o.append("}")
    return "\n".join(o), len(sub_body_fp2(L))


# ---- Fp2 squaring as one subroutine: c0 = (a0 + a1)(a0 - a1), c1 = (2 a0) a1, one reduction each -------------------------
# a0 v0.., a1 vL.. (preserved) -> c0 v2L.., c1 v3L..; s = a0 + a1 v4L.., d = a0 - a1 v5L.., t = 2 a1 v6L..; inputs with
# A <= 2 (L products of magnitude <= 8 2^56 plus L reduction terms per column).
def sub_body_fp2sqr(L):
    A0 = lambda i: "v%d" % i
    A1 = lambda i: "v%d" % (L + i)
    R0 = lambda i: "v%d" % (2 * L + i)
    R1 = lambda i: "v%d" % (3 * L + i)
    S = lambda i: "v%d" % (4 * L + i)
    D = lambda i: "v%d" % (5 * L + i)
    T = lambda i: "v%d" % (6 * L + i)
    P = lambda i: "s%d" % (40 + i)
    INV = "s%d" % (40 + L)
    AC0, LO0 = "v[%d:%d]" % (7 * L, 7 * L + 1), "v%d" % (7 * L)
    AC1, LO1 = "v[%d:%d]" % (7 * L + 2, 7 * L + 3), "v%d" % (7 * L + 2)
    out = []
    for i in range(L):
        out.append("v_add_u32 %s, %s, %s" % (S(i), A0(i), A1(i)))
        out.append("v_sub_u32 %s, %s, %s" % (D(i), A0(i), A1(i)))
        out.append("v_lshlrev_b32 %s, 1, %s" % (T(i), A1(i)))
    f0 = f1 = True
    for k in range(2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            j = k - i
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, S(i), D(j), "0" if f0 else AC0))
            f0 = False
            out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A0(i), T(j), "0" if f1 else AC1))
            f1 = False
        for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(i), P(k - i), AC0))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(i), P(k - i), AC1))
        if k < L:
            out.append("v_mul_lo_u32 %s, %s, %s" % (R0(k), LO0, INV))
            out.append("v_mul_lo_u32 %s, %s, %s" % (R1(k), LO1, INV))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R0(k), R0(k)))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R1(k), R1(k)))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(k), P(0), AC0))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(k), P(0), AC1))
        else:
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R0(k - L), LO0))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R1(k - L), LO1))
        out.append("v_ashrrev_i64 %s, 28, %s" % (AC0, AC0))
        out.append("v_ashrrev_i64 %s, 28, %s" % (AC1, AC1))
    out.append("v_mov_b32 %s, %s" % (R0(L - 1), LO0))
    out.append("v_mov_b32 %s, %s" % (R1(L - 1), LO1))
    out.append("s_setpc_b64 s[34:35]")
    return out


def gen_fp2sqr_call(L):
    NLT = "\\n\\t"
    sym = "gs_fp2sqr28_sub_%d" % L
    o = ['extern "C" __device__ void %s();' % sym]
    body = ["s_branch .Lgs_skipq%d_%%=" % L, ".p2align 8", ".globl %s" % sym, ".type %s,@function" % sym, sym + ":"]
    body += sub_body_fp2sqr(L) + [".Lgs_skipq%d_%%=:" % L]
    o.append('extern "C" __device__ __attribute__((used, noinline)) void gs_fp2sqr28_sub_holder_%d() {' % L)
    o.append('  asm volatile("%s" ::: "memory");' % NLT.join(body))
    o.append("}")
    outs = ", ".join('"={v%d}"(r0[%d])' % (2 * L + i, i) for i in range(L)) + ", " + \
        ", ".join('"={v%d}"(r1[%d])' % (3 * L + i, i) for i in range(L))
    ins = []
    for base, nm in ((0, "a0"), (L, "a1")):
        ins += ['"{v%d}"(%s[%d])' % (base + i, nm, i) for i in range(L)]
    ins += ['"{s%d}"(C::P28[%d])' % (40 + i, i) for i in range(L)] + ['"{s%d}"(C::P28_INV)' % (40 + L)]
    nin = 2 * L + L + 1
    ins.append('"s"((uint64_t)(uintptr_t)&%s)' % sym)
    clob = ['"v%d"' % (4 * L + i) for i in range(3 * L + 4)] + ['"vcc"', '"s34"', '"s35"']
    o.append("template <class C> __device__ __forceinline__ void fp2sqr28_call_%d(int32_t (&r0)[%d], int32_t (&r1)[%d], "
             "const int32_t (&a0)[%d], const int32_t (&a1)[%d]) {" % (L, L, L, L, L))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (2 * L + nin))
    o.append("      : %s" % outs)
    o.append("      : %s" % ", ".join(ins))
    o.append("      : %s);" % ", ".join(clob))
    o.append("}")
    return "\n".join(o), len(sub_body_fp2sqr(L))


# ---- Fp2 dot product of n pairs as one subroutine: c = sum_t a_t b_t, one reduction per output -----------------------
# Pair t: a_t0 v(4tL).., a_t1 v(4tL+L).., b_t0 v(4tL+2L).., b_t1 v(4tL+3L).. (preserved) -> c0 v(4nL).., c1 v(4nL+L)..;
# -a_t1 v(4nL+2L+tL).., two accumulators after them.  Contract: sum_t A_at A_bt <= 4 (2n products of L terms plus L
# reduction terms per column of a signed 64-bit accumulator).  (4n + 2) L^2 multiply-adds for what n separate products
# spend 6n L^2 on: the sparse Miller-line products are six of these with n = 3 and nothing else.
def sub_body_fp2dot(L, n):
    A0 = lambda t, i: "v%d" % (4 * t * L + i)
    A1 = lambda t, i: "v%d" % (4 * t * L + L + i)
    B0 = lambda t, i: "v%d" % (4 * t * L + 2 * L + i)
    B1 = lambda t, i: "v%d" % (4 * t * L + 3 * L + i)
    R0 = lambda i: "v%d" % (4 * n * L + i)
    R1 = lambda i: "v%d" % (4 * n * L + L + i)
    N1 = lambda t, i: "v%d" % (4 * n * L + 2 * L + t * L + i)
    base = 4 * n * L + 2 * L + n * L
    P = lambda i: "s%d" % (40 + i)
    INV = "s%d" % (40 + L)
    AC0, LO0 = "v[%d:%d]" % (base, base + 1), "v%d" % base
    AC1, LO1 = "v[%d:%d]" % (base + 2, base + 3), "v%d" % (base + 2)
    out = []
    for t in range(n):
        for i in range(L):
            out.append("v_sub_u32 %s, 0, %s" % (N1(t, i), A1(t, i)))
    f0 = f1 = True
    for k in range(2 * L - 1):
        for i in range(max(0, k - L + 1), min(k, L - 1) + 1):
            j = k - i
            for t in range(n):
                out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, A0(t, i), B0(t, j), "0" if f0 else AC0))
                f0 = False
                out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A0(t, i), B1(t, j), "0" if f1 else AC1))
                f1 = False
                out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC0, N1(t, i), B1(t, j), AC0))
                out.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (AC1, A1(t, i), B0(t, j), AC1))
        for i in range(max(0, k - L + 1), min(k - 1, L - 1) + 1):
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(i), P(k - i), AC0))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(i), P(k - i), AC1))
        if k < L:
            out.append("v_mul_lo_u32 %s, %s, %s" % (R0(k), LO0, INV))
            out.append("v_mul_lo_u32 %s, %s, %s" % (R1(k), LO1, INV))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R0(k), R0(k)))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R1(k), R1(k)))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC0, R0(k), P(0), AC0))
            out.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (AC1, R1(k), P(0), AC1))
        else:
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R0(k - L), LO0))
            out.append("v_and_b32 %s, 0xfffffff, %s" % (R1(k - L), LO1))
        out.append("v_ashrrev_i64 %s, 28, %s" % (AC0, AC0))
        out.append("v_ashrrev_i64 %s, 28, %s" % (AC1, AC1))
    out.append("v_mov_b32 %s, %s" % (R0(L - 1), LO0))
    out.append("v_mov_b32 %s, %s" % (R1(L - 1), LO1))
    out.append("s_setpc_b64 s[34:35]")
    return out


def gen_fp2dot_call(L, n):
    NLT = "\\n\\t"
    sym = "gs_fp2dot%d_28_sub_%d" % (n, L)
    o = ['extern "C" __device__ void %s();' % sym]
    body = ["s_branch .Lgs_skipd%d_%d_%%=" % (n, L), ".p2align 8", ".globl %s" % sym, ".type %s,@function" % sym, sym + ":"]
    body += sub_body_fp2dot(L, n) + [".Lgs_skipd%d_%d_%%=:" % (n, L)]
    o.append('extern "C" __device__ __attribute__((used, noinline)) void gs_fp2dot%d_28_sub_holder_%d() {' % (n, L))
    o.append('  asm volatile("%s" ::: "memory");' % NLT.join(body))
    o.append("}")
    outs = ", ".join('"={v%d}"(r0[%d])' % (4 * n * L + i, i) for i in range(L)) + ", " + \
        ", ".join('"={v%d}"(r1[%d])' % (4 * n * L + L + i, i) for i in range(L))
    ins = []
    sig = []
    for t in range(n):
        for q, nm in enumerate(("a%d0" % t, "a%d1" % t, "b%d0" % t, "b%d1" % t)):
            ins += ['"{v%d}"(%s[%d])' % (4 * t * L + q * L + i, nm, i) for i in range(L)]
            sig.append("const int32_t (&%s)[%d]" % (nm, L))
    ins += ['"{s%d}"(C::P28[%d])' % (40 + i, i) for i in range(L)] + ['"{s%d}"(C::P28_INV)' % (40 + L)]
    nin = 4 * n * L + L + 1
    ins.append('"s"((uint64_t)(uintptr_t)&%s)' % sym)
    clob = ['"v%d"' % (4 * n * L + 2 * L + i) for i in range(n * L + 4)] + ['"vcc"', '"s34"', '"s35"']
    o.append("template <class C> __device__ __forceinline__ void fp2dot%d_28_call_%d(int32_t (&r0)[%d], int32_t (&r1)[%d], %s) {"
             % (n, L, L, L, ", ".join(sig)))
    o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (2 * L + nin))
    o.append("      : %s" % outs)
    o.append("      : %s" % ", ".join(ins))
    o.append("      : %s);" % ", ".join(clob))
    o.append("}")
    return "\n".join(o), len(sub_body_fp2dot(L, n))


def gen_calls(L):
    NL = "\\n\\t"  # the two escapes as they must appear inside the C string literal
    o = []
    for name in ("mul", "sqr"):
        o.append('extern "C" __device__ void gs_%s28_sub_%d();' % (name, L))
    body = ["s_branch .Lgs_skip%d_%%=" % L]
    for name, sq in (("mul", False), ("sqr", True)):
        sym = "gs_%s28_sub_%d" % (name, L)
        body += [".p2align 8", ".globl %s" % sym, ".type %s,@function" % sym, sym + ":"] + sub_body(L, sq)
    body.append(".Lgs_skip%d_%%=:" % L)
    o.append("// never executed: carries the subroutines' code (their labels are the symbols declared above)")
    o.append('extern "C" __device__ __attribute__((used, noinline)) void gs_fq28_sub_holder_%d() {' % L)
    o.append('  asm volatile("%s" ::: "memory");' % NL.join(body))
    o.append("}")
    for name, sq in (("mul", False), ("sqr", True)):
        outs = ", ".join('"={v%d}"(r[%d])' % (2 * L + i, i) for i in range(L))
        ins = ", ".join('"{v%d}"(a[%d])' % (i, i) for i in range(L))
        nin = L
        if not sq:
            ins += ", " + ", ".join('"{v%d}"(b[%d])' % (L + i, i) for i in range(L))
            nin += L
        ins += ", " + ", ".join('"{s%d}"(C::P28[%d])' % (40 + i, i) for i in range(L)) + ', "{s%d}"(C::P28_INV)' % (40 + L)
        nin += L + 1
        ins += ', "s"((uint64_t)(uintptr_t)&gs_%s28_sub_%d)' % (name, L)
        clob = ['"v%d"' % (3 * L + i) for i in range(3)] + ['"vcc"', '"s34"', '"s35"']
        if sq:
            clob += ['"v%d"' % (L + i) for i in range(L)]
        sig = "const int32_t (&a)[%d]" % L + ("" if sq else ", const int32_t (&b)[%d]" % L)
        o.append("template <class C> __device__ __forceinline__ void %s28_call_%d(int32_t (&r)[%d], %s) {" % (name, L, L, sig))
        o.append('  asm("s_swappc_b64 s[34:35], %%%d"' % (L + nin))
        o.append("      : %s" % outs)
        o.append("      : %s" % ins)
        o.append("      : %s);" % ", ".join(clob))
        o.append("}")
    return "\n".join(o)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    o = ["// GENERATED by gen_mul28_asm.py -- do not edit.  (included inside namespace gs)"]
    for L in (14, 10):
        src, cnt = gen(L)
        o.append("// L = %d: %d instructions" % (L, cnt))
        o.append(src)
        src, cnt = gen_sqr(L)
        o.append("// squaring, L = %d: %d instructions" % (L, cnt))
        o.append(src)
    o.append("#if !defined(GS_NO_ASM_CALL)")
    for L in (14, 10):
        o.append("// ---- subroutine form, L = %d" % L)
        o.append(gen_calls(L))
        src, cnt = gen_fp2_call(L)
        o.append("// Fp2 product (schoolbook, lazy reduction), L = %d: %d instructions" % (L, cnt))
        o.append(src)
        src, cnt = gen_fp2sqr_call(L)
        o.append("// Fp2 squaring, L = %d: %d instructions" % (L, cnt))
        o.append(src)
        src, cnt = gen_fp2dot_call(L, 3)
        o.append("// Fp2 dot product of 3 pairs, L = %d: %d instructions" % (L, cnt))
        o.append(src)
    o.append("#endif")
    with open(os.path.join(here, "gs_mul28_asm.h"), "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote gs_mul28_asm.h")


if __name__ == "__main__":
    main()
