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
    with open(os.path.join(here, "gs_mul28_asm.h"), "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote gs_mul28_asm.h")


if __name__ == "__main__":
    main()
