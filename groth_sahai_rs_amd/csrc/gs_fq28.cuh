// Base field Fq in an UNSATURATED radix-2^28 signed-limb representation, built
// for what gfx950's VALU is good at (measured, profiles/r1/ubench_valu.txt):
//   * v_mad_i64_i32 / v_mad_u64_u32 (32x32 + 64-bit accumulate): ~4 cycles
//   * carry-flag arithmetic (v_add_co / v_addc_co): ALSO ~4 cycles per instruction
//   * plain v_add_u32 / v_and / shifts: ~2 cycles
// A saturated 32-bit-limb Montgomery product needs a carry-flag instruction per
// multiply-accumulate and every modular add/sub is a 36-instruction carry chain.
// With 28-bit limbs up to 14+14 products fit a signed 64-bit accumulator, so a
// Montgomery product is L*L*2 bare mads (+ ~6 ops per column) and field
// add/sub/neg are L independent full-rate adds with NO reduction at all.
//
// Representation: element a is held as the integer  V = a * 2^(28 L) mod p, lazily
// reduced: V may be any representative with |V| < 2^(28L-6), limbs are int32 with
// V = sum v[i] 2^(28 i).  "Normalised" means v[0..L-2] in [0, 2^28 + small) and the
// top limb signed.  mul() returns normalised limbs and a value in (-p/2, 3p/2).
// Contract for mul(a, b): 14 * max|a_i| * max|b_j| + 14 * 2^56 < 2^63, i.e. the
// limb-growth factors satisfy A*B <= 8 (a sum of two normalised values has A = 2);
// callers call norm() where a longer sum feeds a product.  With GS_FQ28_CHECK
// (CPU twin only) every mul asserts the contract.
//
// Boundary form (arkworks, include/gs_amd.h) is canonical saturated Montgomery
// with radix 2^(32N); fq_from_boundary / fq_to_boundary convert with one
// multiplication each, only at kernel inputs/outputs.
#pragma once
#include "gs_field.cuh"
#if defined(GS_FQ28_CHECK)
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#endif

namespace gs {

constexpr int32_t M28 = 0x0FFFFFFF;

#if defined(GS_FQ28_CHECK)
typedef int64_t limb_t;  // CPU-twin debug build: wide limbs so that int32 overflow is detectable
#define GS_CHK_LIMBS(r)                                                                        \
  for (int i_ = 0; i_ < C::L; i_++)                                                            \
    if ((r).v[i_] >= ((int64_t)1 << 31) || (r).v[i_] < -((int64_t)1 << 31)) {                 \
      fprintf(stderr, "Fq28 limb overflow (|limb| >= 2^31) at %s:%d\n", __FILE__, __LINE__);   \
      abort();                                                                                 \
    }
#else
typedef int32_t limb_t;
#define GS_CHK_LIMBS(r)
#endif

template <class C> struct Fq28 {
  limb_t v[C::L];
};

#if defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM)
#include "gs_mul28_asm.h"
#if !defined(GS_NO_ASM_CALL) && !defined(GS_NO_POINT_ASM)
// whole G1 point operations as subroutines with their own register allocation (gen_pointops_asm.py; used by gs_curve.cuh)
#define GS_POINT_ASM 1
#if !defined(GS_NO_POINT_ASM_G2) && !defined(GS_POINT_ASM_G2_STRAIGHT) && !defined(GS_POINT_ASM_G2)
#define GS_POINT_ASM_G2 1  // G2 too: the compact form (generated data movement around the shared Fp2 subroutines)
#endif
template <class C> constexpr uint64_t pinv56();  // (below; the G2 subroutines take it in s[60:61])
#include "gs_pointops_asm.h"
#endif
#endif

// ---- lazy linear operations (no carries, no reduction) -------------------------
template <class C> GS_HD Fq28<C> add(const Fq28<C>& a, const Fq28<C>& b) {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = a.v[i] + b.v[i];
  GS_CHK_LIMBS(r)
  return r;
}
template <class C> GS_HD Fq28<C> sub(const Fq28<C>& a, const Fq28<C>& b) {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = a.v[i] - b.v[i];
  GS_CHK_LIMBS(r)
  return r;
}
template <class C> GS_HD Fq28<C> neg(const Fq28<C>& a) {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = -a.v[i];
  return r;
}
template <class C> GS_HD Fq28<C> dbl(const Fq28<C>& a) {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = a.v[i] * 2;
  GS_CHK_LIMBS(r)
  return r;
}
template <class C> GS_HD Fq28<C> mul_small(const Fq28<C>& a, int k) {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = a.v[i] * k;
  GS_CHK_LIMBS(r)
  return r;
}
template <class C> GS_HD Fq28<C> select(bool c, const Fq28<C>& a, const Fq28<C>& b) {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
template <class C> GS_HD Fq28<C> fq_zero() {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = 0;
  return r;
}
template <class C> GS_HD Fq28<C> fq_one() {
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = C::ONE28[i];
  return r;
}
// exact all-limbs-zero test: used for identity flags (identities are always
// stored as exact zeros and exact zero is absorbing under mul/dbl)
template <class C> GS_HD bool is_zero_limbs(const Fq28<C>& a) {
  limb_t o = 0;
#pragma unroll
  for (int i = 0; i < C::L; i++) o |= a.v[i];
  return o == 0;
}

// one parallel carry round: limbs 0..L-2 back to [0, 2^28 + small), top signed
template <class C> GS_HD Fq28<C> norm(const Fq28<C>& a) {
  Fq28<C> r;
  r.v[0] = a.v[0] & M28;
#pragma unroll
  for (int i = 1; i < C::L - 1; i++) r.v[i] = (a.v[i] & M28) + (a.v[i - 1] >> 28);
  r.v[C::L - 1] = a.v[C::L - 1] + (a.v[C::L - 2] >> 28);
  return r;
}
// sequential carry propagation: unique limb vector for the integer V
template <class C> GS_HD Fq28<C> norm_full(const Fq28<C>& a) {
  Fq28<C> r;
  limb_t c = 0;
#pragma unroll
  for (int i = 0; i < C::L - 1; i++) {
    limb_t t = a.v[i] + c;
    r.v[i] = t & M28;
    c = t >> 28;
  }
  r.v[C::L - 1] = a.v[C::L - 1] + c;
  return r;
}

// value reduction: subtract round(V/p) * p (quotient estimated in f32 from the two top
// limbs).  Needed only where a value is fed back LINEARLY (cyclotomic squaring:
// z <- 3t - 2z doubles |V| per step); everywhere else multiplication contracts values.
// Input: weakly normalised limbs, |V| < 2^20 p.  Output: unique limbs, |V| <= ~p.
template <class C> GS_HD Fq28<C> vreduce(const Fq28<C>& a) {
  constexpr int L = C::L;
  float v = (float)a.v[L - 1] * 268435456.0f + (float)a.v[L - 2];
  int32_t k = (int32_t)__builtin_rintf(v * C::INV_PTOP2);
  Fq28<C> r;
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < L - 1; i++) {
    acc += (int64_t)a.v[i] - (int64_t)k * C::P28[i];
    r.v[i] = (limb_t)(((uint32_t)acc) & (uint32_t)M28);
    acc >>= 28;
  }
  r.v[L - 1] = (limb_t)((int64_t)a.v[L - 1] - (int64_t)k * C::P28[L - 1] + acc);
  return r;
}

// ---- Montgomery product ----------------------------------------------------------
template <class C, class T> GS_HD void mul28_generic(T* r, const T* a, const T* b) {
  constexpr int L = C::L;
  uint32_t m[L];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j < 0 || j >= L) continue;
      acc += (int64_t)a[i] * b[j];
      if (j >= 1 && i < k) acc += (int64_t)(int32_t)m[i] * C::P28[j];
    }
    if (k < L) {
      m[k] = ((((uint32_t)acc) & (uint32_t)M28) * C::P28_INV) & (uint32_t)M28;
      acc += (int64_t)(int32_t)m[k] * C::P28[0];
      acc >>= 28;
    } else {
      r[k - L] = (T)(((uint32_t)acc) & (uint32_t)M28);
      acc >>= 28;
    }
  }
  r[L - 1] = (T)acc;
}

#if defined(GS_FQ28_CHECK)
// CPU twin only: number of Fq multiplications executed (feeds the ALU roofline of bench.py through
// tools/count_fq_muls.py)
inline std::atomic<long>& fq28_mul_counter() {
  static std::atomic<long> n{0};
  return n;
}
// ... and the multiply-add instructions the DEVICE kernels execute for them (static counts of gs_mul28_asm.h: product
// 2 L^2, squaring L (L + 1) / 2 + L^2, Fp2 product 6 L^2, Fp2 squaring 4 L^2, three-term Fp2 dot product 14 L^2): the
// "executed" side of bench.py's ALU accounting, where the counter above credits a squaring as a full product
inline std::atomic<long>& fq28_mad_counter() {
  static std::atomic<long> n{0};
  return n;
}
template <class C> inline void fq28_check(const Fq28<C>& a, const Fq28<C>& b) {
  fq28_mul_counter().fetch_add(1, std::memory_order_relaxed);
  fq28_mad_counter().fetch_add(2 * C::L * C::L, std::memory_order_relaxed);
  int64_t ma = 0, mb = 0;
  for (int i = 0; i < C::L; i++) {
    int64_t x = a.v[i] < 0 ? -(int64_t)a.v[i] : a.v[i], y = b.v[i] < 0 ? -(int64_t)b.v[i] : b.v[i];
    if (x > ma) ma = x;
    if (y > mb) mb = y;
  }
  // sum of L products + L reduction terms must stay below 2^63
  __int128 worst = (__int128)C::L * ma * mb + (__int128)C::L * ((__int128)1 << 56);
  int64_t ta = a.v[C::L - 1] < 0 ? -(int64_t)a.v[C::L - 1] : a.v[C::L - 1];
  int64_t tb = b.v[C::L - 1] < 0 ? -(int64_t)b.v[C::L - 1] : b.v[C::L - 1];
  // value sanity: |V| < 2^(28L - 2); mul contracts values (|ab|/R + p), lazy chains stay far below this,
  // only linear feedback (cyclotomic squaring) needs vreduce()
  bool val_ok = ta < (1 << 26) && tb < (1 << 26);
  if (worst >= ((__int128)1 << 63) || !val_ok) {
    fprintf(stderr, "Fq28 mul contract violated: max|a_i|=%lld max|b_j|=%lld top %lld %lld\n", (long long)ma,
            (long long)mb, (long long)ta, (long long)tb);
    abort();
  }
}
#endif

// register-passing wrapper: the out-of-line body takes/returns ext-vectors so that
// operands stay in VGPRs across the call (same trick as gs_field.cuh)
typedef int32_t i32x16 __attribute__((ext_vector_type(16)));
template <class C> GS_HD i32x16 mul28_body(i32x16 a, i32x16 b) {
  int32_t av[C::L], bv[C::L], rv[C::L];
#pragma unroll
  for (int i = 0; i < C::L; i++) {
    av[i] = a[i];
    bv[i] = b[i];
  }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM)
  if constexpr (C::L == 14)
    mul28_asm_14<C>(rv, av, bv);
  else
    mul28_asm_10<C>(rv, av, bv);
#else
  mul28_generic<C, int32_t>(rv, av, bv);
#endif
  i32x16 r = 0;
#pragma unroll
  for (int i = 0; i < C::L; i++) r[i] = rv[i];
  return r;
}

template <class C> GS_HD_NOINLINE i32x16 mul28_vec(i32x16 a, i32x16 b) { return mul28_body<C>(a, b); }

template <class C> GS_HD Fq28<C> mul(const Fq28<C>& a, const Fq28<C>& b) {
#if defined(GS_FQ28_CHECK)
  fq28_check(a, b);
  Fq28<C> rr;
  mul28_generic<C, limb_t>(rr.v, a.v, b.v);
  return rr;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM) && !defined(GS_NO_ASM_CALL)
  // the multiplier as a SUBROUTINE reached from an asm statement with fixed operand registers (gs_mul28_asm.h,
  // "subroutine form"): the hardware call without a C++ call's s_waitcnt vmcnt(0) at the callee's entry
  Fq28<C> r;
  if constexpr (C::L == 14)
    mul28_call_14<C>(r.v, a.v, b.v);
  else
    mul28_call_10<C>(r.v, a.v, b.v);
  return r;
#else
  i32x16 av = 0, bv = 0;
#pragma unroll
  for (int i = 0; i < C::L; i++) {
    av[i] = a.v[i];
    bv[i] = b.v[i];
  }
  i32x16 rv = mul28_vec<C>(av, bv);
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = rv[i];
  return r;
#endif
}
// squaring: its own out-of-line body on the device (L(L+1)/2 product mads against the doubled operand instead of
// L^2, and one operand to marshal instead of two); same result and contract as mul(a, a)
template <class C> GS_HD i32x16 sqr28_body(i32x16 a) {
  int32_t av[C::L], dv[C::L], rv[C::L];
#pragma unroll
  for (int i = 0; i < C::L; i++) {
    av[i] = a[i];
    dv[i] = a[i] * 2;
  }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM)
  if constexpr (C::L == 14)
    sqr28_asm_14<C>(rv, av, dv);
  else
    sqr28_asm_10<C>(rv, av, dv);
#else
  (void)dv;
  mul28_generic<C, int32_t>(rv, av, av);
#endif
  i32x16 r = 0;
#pragma unroll
  for (int i = 0; i < C::L; i++) r[i] = rv[i];
  return r;
}
template <class C> GS_HD_NOINLINE i32x16 sqr28_vec(i32x16 a) { return sqr28_body<C>(a); }
template <class C> GS_HD Fq28<C> sqr(const Fq28<C>& a) {
#if defined(GS_FQ28_CHECK)
  fq28_mad_counter().fetch_sub(C::L * (C::L - 1) / 2, std::memory_order_relaxed);  // the device squares with fewer mads
  return mul(a, a);
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM) && !defined(GS_NO_ASM_CALL)
  Fq28<C> r;
  if constexpr (C::L == 14)
    sqr28_call_14<C>(r.v, a.v);
  else
    sqr28_call_10<C>(r.v, a.v);
  return r;
#else
  i32x16 av = 0;
#pragma unroll
  for (int i = 0; i < C::L; i++) av[i] = a.v[i];
  i32x16 rv = sqr28_vec<C>(av);
  Fq28<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) r.v[i] = rv[i];
  return r;
#endif
}

// ---- Fp2 product kernel: schoolbook with lazy reduction ------------------------------------------------------------
//   c0 = a0 b0 - a1 b1,  c1 = a0 b1 + a1 b0 : two product-scanning sums per accumulator, ONE Montgomery reduction per
// output (gs_mul28_asm.h, "Fp2 product").  Same mads as Karatsuba's three full products, none of its five lazy
// additions and no carry round on the result; both outputs normalised.  Contract: A_a * A_b <= 4 (2L products + L
// reduction terms per column of a signed 64-bit accumulator); the CPU twin asserts it per column.
template <class C, class T> GS_HD void fp2mul28_generic(T* r0, T* r1, const T* a0, const T* a1, const T* b0, const T* b1) {
  constexpr int L = C::L;
  uint32_t m0[L], m1[L];
#if defined(GS_FQ28_CHECK)
  __int128 acc0 = 0, acc1 = 0;
#else
  int64_t acc0 = 0, acc1 = 0;
#endif
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j < 0 || j >= L) continue;
      acc0 += (int64_t)a0[i] * b0[j] - (int64_t)a1[i] * b1[j];
      acc1 += (int64_t)a0[i] * b1[j] + (int64_t)a1[i] * b0[j];
      if (j >= 1 && i < k) {
        acc0 += (int64_t)(int32_t)m0[i] * C::P28[j];
        acc1 += (int64_t)(int32_t)m1[i] * C::P28[j];
      }
    }
#if defined(GS_FQ28_CHECK)
    {
      const __int128 lim = (__int128)1 << 63;
      if (acc0 >= lim || acc0 < -lim || acc1 >= lim || acc1 < -lim) {
        fprintf(stderr, "Fp2 product: column accumulator leaves the signed 64-bit range (A_a * A_b > 4)\n");
        abort();
      }
    }
#endif
    if (k < L) {
      m0[k] = ((((uint32_t)acc0) & (uint32_t)M28) * C::P28_INV) & (uint32_t)M28;
      m1[k] = ((((uint32_t)acc1) & (uint32_t)M28) * C::P28_INV) & (uint32_t)M28;
      acc0 += (int64_t)(int32_t)m0[k] * C::P28[0];
      acc1 += (int64_t)(int32_t)m1[k] * C::P28[0];
    } else {
      r0[k - L] = (T)(((uint32_t)acc0) & (uint32_t)M28);
      r1[k - L] = (T)(((uint32_t)acc1) & (uint32_t)M28);
    }
    acc0 >>= 28;
    acc1 >>= 28;
  }
  r0[L - 1] = (T)acc0;
  r1[L - 1] = (T)acc1;
}
template <class C> GS_HD void fp2mul28(Fq28<C>& r0, Fq28<C>& r1, const Fq28<C>& a0, const Fq28<C>& a1, const Fq28<C>& b0,
                                       const Fq28<C>& b1) {
#if defined(GS_FQ28_CHECK)
  fq28_mul_counter().fetch_add(3, std::memory_order_relaxed);  // counted as the three products it replaces
  fq28_mad_counter().fetch_add(6 * C::L * C::L, std::memory_order_relaxed);
  for (const Fq28<C>* x : {&a0, &a1, &b0, &b1}) {
    int64_t t = x->v[C::L - 1] < 0 ? -(int64_t)x->v[C::L - 1] : x->v[C::L - 1];
    if (t >= (1 << 26)) {
      fprintf(stderr, "Fp2 product: operand value out of range (top limb %lld)\n", (long long)t);
      abort();
    }
  }
  fp2mul28_generic<C, limb_t>(r0.v, r1.v, a0.v, a1.v, b0.v, b1.v);
  GS_CHK_LIMBS(r0)
  GS_CHK_LIMBS(r1)
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM) && !defined(GS_NO_ASM_CALL)
  if constexpr (C::L == 14)
    fp2mul28_call_14<C>(r0.v, r1.v, a0.v, a1.v, b0.v, b1.v);
  else
    fp2mul28_call_10<C>(r0.v, r1.v, a0.v, a1.v, b0.v, b1.v);
#else
  fp2mul28_generic<C, int32_t>(r0.v, r1.v, a0.v, a1.v, b0.v, b1.v);
#endif
}

// ---- Fp2 dot product kernel: c = a0 b0 + a1 b1 + a2 b2 in Fp2, ONE Montgomery reduction per output ----------------
// (gs_mul28_asm.h, "Fp2 dot product of 3 pairs").  14 L^2 multiply-adds for what three separate products spend
// 18 L^2 on, and the sum needs no lazy additions or carry round afterwards: the sparse Miller-line products are six of
// these (gs_tower.cuh).  Contract: sum_t A_at A_bt <= 4; the CPU twin asserts it per column.  a / b: three (re, im).
template <class C, class T> GS_HD void fp2dot3_28_generic(T* r0, T* r1, const T* const* a, const T* const* b) {
  constexpr int L = C::L;
  uint32_t m0[L], m1[L];
#if defined(GS_FQ28_CHECK)
  __int128 acc0 = 0, acc1 = 0;
#else
  int64_t acc0 = 0, acc1 = 0;
#endif
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j < 0 || j >= L) continue;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        acc0 += (int64_t)a[2 * t][i] * b[2 * t][j] - (int64_t)a[2 * t + 1][i] * b[2 * t + 1][j];
        acc1 += (int64_t)a[2 * t][i] * b[2 * t + 1][j] + (int64_t)a[2 * t + 1][i] * b[2 * t][j];
      }
      if (j >= 1 && i < k) {
        acc0 += (int64_t)(int32_t)m0[i] * C::P28[j];
        acc1 += (int64_t)(int32_t)m1[i] * C::P28[j];
      }
    }
#if defined(GS_FQ28_CHECK)
    {
      const __int128 lim = (__int128)1 << 63;
      if (acc0 >= lim || acc0 < -lim || acc1 >= lim || acc1 < -lim) {
        fprintf(stderr, "Fp2 dot product: column accumulator leaves the signed 64-bit range (sum A_a A_b > 4)\n");
        abort();
      }
    }
#endif
    if (k < L) {
      m0[k] = ((((uint32_t)acc0) & (uint32_t)M28) * C::P28_INV) & (uint32_t)M28;
      m1[k] = ((((uint32_t)acc1) & (uint32_t)M28) * C::P28_INV) & (uint32_t)M28;
      acc0 += (int64_t)(int32_t)m0[k] * C::P28[0];
      acc1 += (int64_t)(int32_t)m1[k] * C::P28[0];
    } else {
      r0[k - L] = (T)(((uint32_t)acc0) & (uint32_t)M28);
      r1[k - L] = (T)(((uint32_t)acc1) & (uint32_t)M28);
    }
    acc0 >>= 28;
    acc1 >>= 28;
  }
  r0[L - 1] = (T)acc0;
  r1[L - 1] = (T)acc1;
}
// r = a0 b0 + a1 b1 + a2 b2 ; every operand as (re, im)
template <class C>
GS_HD void fp2dot3_28(Fq28<C>& r0, Fq28<C>& r1, const Fq28<C>& a00, const Fq28<C>& a01, const Fq28<C>& b00,
                      const Fq28<C>& b01, const Fq28<C>& a10, const Fq28<C>& a11, const Fq28<C>& b10, const Fq28<C>& b11,
                      const Fq28<C>& a20, const Fq28<C>& a21, const Fq28<C>& b20, const Fq28<C>& b21) {
#if defined(GS_FQ28_CHECK)
  fq28_mul_counter().fetch_add(7, std::memory_order_relaxed);  // 14 L^2 multiply-adds = 7 products' worth
  fq28_mad_counter().fetch_add(14 * C::L * C::L, std::memory_order_relaxed);
  for (const Fq28<C>* x : {&a00, &a01, &b00, &b01, &a10, &a11, &b10, &b11, &a20, &a21, &b20, &b21}) {
    int64_t t = x->v[C::L - 1] < 0 ? -(int64_t)x->v[C::L - 1] : x->v[C::L - 1];
    if (t >= (1 << 26)) {
      fprintf(stderr, "Fp2 dot product: operand value out of range (top limb %lld)\n", (long long)t);
      abort();
    }
  }
  const limb_t* a[6] = {a00.v, a01.v, a10.v, a11.v, a20.v, a21.v};
  const limb_t* b[6] = {b00.v, b01.v, b10.v, b11.v, b20.v, b21.v};
  fp2dot3_28_generic<C, limb_t>(r0.v, r1.v, a, b);
  GS_CHK_LIMBS(r0)
  GS_CHK_LIMBS(r1)
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM) && !defined(GS_NO_ASM_CALL)
  if constexpr (C::L == 14)
    fp2dot3_28_call_14<C>(r0.v, r1.v, a00.v, a01.v, b00.v, b01.v, a10.v, a11.v, b10.v, b11.v, a20.v, a21.v, b20.v, b21.v);
  else
    fp2dot3_28_call_10<C>(r0.v, r1.v, a00.v, a01.v, b00.v, b01.v, a10.v, a11.v, b10.v, b11.v, a20.v, a21.v, b20.v, b21.v);
#else
  const int32_t* a[6] = {a00.v, a01.v, a10.v, a11.v, a20.v, a21.v};
  const int32_t* b[6] = {b00.v, b01.v, b10.v, b11.v, b20.v, b21.v};
  fp2dot3_28_generic<C, int32_t>(r0.v, r1.v, a, b);
#endif
}

// Fp2 squaring kernel: c0 = (a0 + a1)(a0 - a1), c1 = (2 a0) a1 with one reduction each; inputs with A <= 2
template <class C, class T> GS_HD void fp2sqr28_generic(T* r0, T* r1, const T* a0, const T* a1) {
  constexpr int L = C::L;
  uint32_t m0[L], m1[L];
  int64_t s[L], d[L], t[L];
#pragma unroll
  for (int i = 0; i < L; i++) {
    s[i] = (int64_t)a0[i] + a1[i];
    d[i] = (int64_t)a0[i] - a1[i];
    t[i] = (int64_t)a1[i] * 2;
#if defined(GS_FQ28_CHECK)
    const int64_t lim = (int64_t)1 << 31;
    if (s[i] >= lim || s[i] < -lim || d[i] >= lim || d[i] < -lim || t[i] >= lim || t[i] < -lim) {
      fprintf(stderr, "Fp2 squaring: operand sums leave the 32-bit range\n");
      abort();
    }
#endif
  }
#if defined(GS_FQ28_CHECK)
  __int128 acc0 = 0, acc1 = 0;
#else
  int64_t acc0 = 0, acc1 = 0;
#endif
#pragma unroll
  for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j < 0 || j >= L) continue;
      acc0 += s[i] * d[j];
      acc1 += (int64_t)a0[i] * t[j];
      if (j >= 1 && i < k) {
        acc0 += (int64_t)(int32_t)m0[i] * C::P28[j];
        acc1 += (int64_t)(int32_t)m1[i] * C::P28[j];
      }
    }
#if defined(GS_FQ28_CHECK)
    {
      const __int128 lim = (__int128)1 << 63;
      if (acc0 >= lim || acc0 < -lim || acc1 >= lim || acc1 < -lim) {
        fprintf(stderr, "Fp2 squaring: column accumulator leaves the signed 64-bit range (A > 2)\n");
        abort();
      }
    }
#endif
    if (k < L) {
      m0[k] = ((((uint32_t)acc0) & (uint32_t)M28) * C::P28_INV) & (uint32_t)M28;
      m1[k] = ((((uint32_t)acc1) & (uint32_t)M28) * C::P28_INV) & (uint32_t)M28;
      acc0 += (int64_t)(int32_t)m0[k] * C::P28[0];
      acc1 += (int64_t)(int32_t)m1[k] * C::P28[0];
    } else {
      r0[k - L] = (T)(((uint32_t)acc0) & (uint32_t)M28);
      r1[k - L] = (T)(((uint32_t)acc1) & (uint32_t)M28);
    }
    acc0 >>= 28;
    acc1 >>= 28;
  }
  r0[L - 1] = (T)acc0;
  r1[L - 1] = (T)acc1;
}
template <class C> GS_HD void fp2sqr28(Fq28<C>& r0, Fq28<C>& r1, const Fq28<C>& a0, const Fq28<C>& a1) {
#if defined(GS_FQ28_CHECK)
  fq28_mul_counter().fetch_add(2, std::memory_order_relaxed);  // counted as the two products it replaces
  fq28_mad_counter().fetch_add(4 * C::L * C::L, std::memory_order_relaxed);
  for (const Fq28<C>* x : {&a0, &a1}) {
    int64_t t = x->v[C::L - 1] < 0 ? -(int64_t)x->v[C::L - 1] : x->v[C::L - 1];
    if (t >= (1 << 26)) {
      fprintf(stderr, "Fp2 squaring: operand value out of range (top limb %lld)\n", (long long)t);
      abort();
    }
  }
  fp2sqr28_generic<C, limb_t>(r0.v, r1.v, a0.v, a1.v);
  GS_CHK_LIMBS(r0)
  GS_CHK_LIMBS(r1)
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM) && !defined(GS_NO_ASM_CALL)
  if constexpr (C::L == 14)
    fp2sqr28_call_14<C>(r0.v, r1.v, a0.v, a1.v);
  else
    fp2sqr28_call_10<C>(r0.v, r1.v, a0.v, a1.v);
#else
  fp2sqr28_generic<C, int32_t>(r0.v, r1.v, a0.v, a1.v);
#endif
}

// ---- tests modulo p --------------------------------------------------------------
// robust a == 0 (mod p) for any lazily reduced a within the mul contract: one
// multiplication by 1 brings the value into (-p/2, 3p/2) with unique limbs, where
// the only multiples of p are 0 and p.
template <class C> GS_HD_NOINLINE bool is_zero_slow(const Fq28<C>& a) {
  Fq28<C> t = norm_full(mul(norm(a), fq_one<C>()));
  limb_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < C::L; i++) {
    z |= t.v[i];
    e |= t.v[i] ^ C::P28[i];
  }
  return z == 0 || e == 0;
}
// p^-1 mod 2^56 from the two low limbs of p (Newton: x <- x (2 - p x) doubles the valid bits)
template <class C> constexpr uint64_t pinv56() {
  const uint64_t p = (uint64_t)(uint32_t)C::P28[0] | ((uint64_t)(uint32_t)C::P28[1] << 28);
  uint64_t x = p;  // p * p = 1 mod 8
  for (int i = 0; i < 6; i++) x *= 2 - p * x;
  return x & (((uint64_t)1 << 56) - 1);
}
// the cheap "cannot be 0 mod p" filter on its own: false = certainly not a multiple of p.  V = k p with |k| < 2^20
// (every lazily reduced value within the mul contract) forces (V mod 2^56) p^-1 mod 2^56 to be the small signed integer
// k.  Only limbs 0 and 1 enter (V mod 2^56 needs no carry propagation), and a non-multiple passes with probability
// 2^-35.  (Rounds 1-3 filtered on 28 bits of the fully carried value: a non-multiple passed with probability 2^-7 PER
// LANE, i.e. in 39 % of the waves, and every point addition asks this about its H.)
template <class C> GS_HD bool maybe_zero_limbs01(limb_t v0, limb_t v1) {
  const uint64_t m56 = ((uint64_t)1 << 56) - 1;
  uint64_t V = (uint64_t)(int64_t)v0 + ((uint64_t)(int64_t)v1 << 28);
  uint64_t k = (V * pinv56<C>()) & m56;
  int64_t ks = (int64_t)(k << 8) >> 8;  // sign-extend 56 bits
  return !(ks > ((int64_t)1 << 20) || ks < -((int64_t)1 << 20));
}
template <class C> GS_HD bool is_zero(const Fq28<C>& a) {
  if (!maybe_zero_limbs01<C>(a.v[0], a.v[1])) return false;
  return is_zero_slow(a);  // the robust test decides
}
template <class C> GS_HD bool eq(const Fq28<C>& a, const Fq28<C>& b) { return is_zero(sub(a, b)); }

// ---- inversion --------------------------------------------------------------------------------------------------
// Round 4: constant-time "safegcd" (Bernstein-Yang divsteps, the half-delta variant as in libsecp256k1's modinv32) on
// signed 28-bit limbs, instead of Fermat's a^(p-2) (380 squarings + ~100 products = ~190 k instructions per lane).
// INV_DIVSTEPS divsteps suffice for a p of this size; they run in groups of 28 on the low words of (f, g) -- 17
// branch-free operations each -- and every group is applied to the full-length (f, g) and (d, e) as one 2 x 2 matrix:
// ~23 k instructions per inversion, uniform control flow (every lane runs the same fixed number of groups).
// Every reduction kernel (one inversion per lane of up to 8 outputs), the final exponentiation's easy part and the
// square-root / decode paths sit behind this; at small batches an inversion is a serial 0.3-0.4 ms each.
//   value in:  V = a R (any lazily reduced representative);  x = V mod p in [0, p)
//   safegcd:   y = x^-1 mod p = a^-1 R^-1;   out = y R^3 R^-1 = a^-1 R   (one product by RRR28 = R^3 mod p)
// inv(0) = 0 (f stays p, d stays 0).  -DGS_INV_FERMAT keeps the exponentiation.
template <class C> GS_HD void inv28_divsteps(int32_t& zeta, uint32_t f0, uint32_t g0, int32_t (&t)[4]) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 28; i++) {
    uint32_t m1 = (uint32_t)(zeta >> 31);  // zeta < 0
    uint32_t m2 = 0u - (g & 1u);           // g odd
    uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;  // conditionally negated f, u, v
    g += x & m2;
    q += y & m2;
    r += z & m2;
    m1 &= m2;
    zeta = (int32_t)((uint32_t)zeta ^ m1) - 1;  // -zeta - 2 or zeta - 1
    f += g & m1;
    u += q & m1;
    v += r & m1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t[0] = (int32_t)u, t[1] = (int32_t)v, t[2] = (int32_t)q, t[3] = (int32_t)r;
}
template <class C> GS_HD_NOINLINE void inv28_safegcd(limb_t* out, const limb_t* in) {
  constexpr int L = C::L;
  constexpr int GROUPS = (C::INV_DIVSTEPS + 1 + 27) / 28;
  constexpr uint32_t PINV = (0u - C::P28_INV) & (uint32_t)M28;  // p^-1 mod 2^28
  // x = the canonical representative of the input value
  Fq28<C> a;
  for (int i = 0; i < L; i++) a.v[i] = in[i];
  Fq28<C> tt = norm_full(mul(a, fq_one<C>()));  // value in (-p/2, 3p/2), unique limbs
  Fq28<C> pp;
  for (int i = 0; i < L; i++) pp.v[i] = C::P28[i];
  Fq28<C> plus = norm_full(add(tt, pp)), minus = norm_full(sub(tt, pp));
  Fq28<C> cx = tt.v[L - 1] < 0 ? plus : (minus.v[L - 1] >= 0 ? minus : tt);
  int32_t f[L], g[L], d[L], e[L];
  for (int i = 0; i < L; i++) {
    f[i] = C::P28[i];
    g[i] = (int32_t)cx.v[i];
    d[i] = 0;
    e[i] = i == 0 ? 1 : 0;
  }
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < GROUPS; it++) {
    int32_t t[4];
    inv28_divsteps<C>(zeta, (uint32_t)f[0] | ((uint32_t)f[1] << 28), (uint32_t)g[0] | ((uint32_t)g[1] << 28), t);
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    {  // (d, e) <- t (d, e) / 2^28 mod p, kept in (-2p, p)
      const int32_t sd = d[L - 1] >> 31, se = e[L - 1] >> 31;
      int32_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);
      int64_t cd = u * d[0] + v * e[0], ce = q * d[0] + r * e[0];
      md -= (int32_t)((PINV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M28);
      me -= (int32_t)((PINV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M28);
      cd += (int64_t)C::P28[0] * md;
      ce += (int64_t)C::P28[0] * me;
      cd >>= 28;
      ce >>= 28;
#pragma unroll
      for (int i = 1; i < L; i++) {
        cd += u * d[i] + v * e[i] + (int64_t)C::P28[i] * md;
        ce += q * d[i] + r * e[i] + (int64_t)C::P28[i] * me;
        d[i - 1] = (int32_t)cd & M28;
        e[i - 1] = (int32_t)ce & M28;
        cd >>= 28;
        ce >>= 28;
      }
      d[L - 1] = (int32_t)cd;
      e[L - 1] = (int32_t)ce;
    }
    {  // (f, g) <- t (f, g) / 2^28 (exact)
      int64_t cf = u * f[0] + v * g[0], cg = q * f[0] + r * g[0];
      cf >>= 28;
      cg >>= 28;
#pragma unroll
      for (int i = 1; i < L; i++) {
        cf += u * f[i] + v * g[i];
        cg += q * f[i] + r * g[i];
        f[i - 1] = (int32_t)cf & M28;
        g[i - 1] = (int32_t)cg & M28;
        cf >>= 28;
        cg >>= 28;
      }
      f[L - 1] = (int32_t)cf;
      g[L - 1] = (int32_t)cg;
    }
  }
  // g = 0, f = +-1 (or +-p for x = 0): y = sign(f) d, brought from (-2p, p) into a lazily reduced internal value
  Fq28<C> y;
  const bool fneg = f[L - 1] < 0;
  for (int i = 0; i < L; i++) y.v[i] = fneg ? -d[i] : d[i];
  Fq28<C> k;
  for (int i = 0; i < L; i++) k.v[i] = C::RRR28[i];
  Fq28<C> res = mul(norm(y), k);  // |y| < 2p: far inside the mul contract
  for (int i = 0; i < L; i++) out[i] = res.v[i];
}

// a^(p-2); inv(0) = 0.  4-bit fixed window over the constant exponent.
template <class C> GS_HD_NOINLINE void inv28_fermat(limb_t* out, const limb_t* in) {
  Fq28<C> a;
  for (int i = 0; i < C::L; i++) a.v[i] = in[i];
  Fq28<C> tab[16];
  tab[0] = fq_one<C>();
  tab[1] = a;
  for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
  // exponent p - 2 from the saturated constant C::P (32-bit limbs)
  Fq28<C> r = fq_one<C>();
  bool started = false;
  for (int w = C::N * 8 - 1; w >= 0; w--) {
    int limb = w >> 3, sh = (w & 7) * 4;
    uint32_t e = C::P[limb];
    if (limb == 0) e -= 2;
    uint32_t dgt = (e >> sh) & 15u;
    if (started) {
      r = sqr(r);
      r = sqr(r);
      r = sqr(r);
      r = sqr(r);
    }
    if (dgt) {
      r = started ? mul(r, tab[dgt]) : tab[dgt];
      started = true;
    }
  }
  for (int i = 0; i < C::L; i++) out[i] = r.v[i];
}
template <class C> GS_HD void inv28_raw(limb_t* out, const limb_t* in) {
#if defined(GS_INV_FERMAT)
  inv28_fermat<C>(out, in);
#else
  inv28_safegcd<C>(out, in);
#endif
}
template <class C> GS_HD Fq28<C> inv(const Fq28<C>& a) {
  Fq28<C> n = norm(a), r;
  inv28_raw<C>(r.v, n.v);
  return r;
}

// ---- boundary conversion ---------------------------------------------------------
// canonical saturated limbs (N x u32, value < p) -> internal
// (out of line: kernels call these at their edges only, and a kernel body with the multiplier's fixed-register
// asm statements inlined into it crashes this LLVM's machine scheduler)
template <class C> GS_HD_NOINLINE Fq28<C> fq_from_boundary(const uint32_t* w) {
  Fq28<C> t;
#pragma unroll
  for (int i = 0; i < C::L; i++) {
    int bit = 28 * i;
    int lo = bit >> 5, sh = bit & 31;
    uint64_t x = 0;
    if (lo < C::N) x = w[lo];
    if (lo + 1 < C::N) x |= (uint64_t)w[lo + 1] << 32;
    t.v[i] = (limb_t)((uint32_t)(x >> sh) & (uint32_t)M28);
  }
  Fq28<C> k;
#pragma unroll
  for (int i = 0; i < C::L; i++) k.v[i] = C::K_IN28[i];
  return mul(t, k);
}
// internal -> canonical saturated limbs
template <class C> GS_HD_NOINLINE void fq_to_boundary(uint32_t* w, const Fq28<C>& a) {
  Fq28<C> k;
#pragma unroll
  for (int i = 0; i < C::L; i++) k.v[i] = C::K_OUT28[i];
  Fq28<C> t = norm_full(mul(norm(a), k));  // value in (-p/2, 3p/2)
  // bring into [0, p)
  Fq28<C> p;
#pragma unroll
  for (int i = 0; i < C::L; i++) p.v[i] = C::P28[i];
  Fq28<C> plus = norm_full(add(t, p)), minus = norm_full(sub(t, p));
  bool negv = t.v[C::L - 1] < 0;
  bool big = minus.v[C::L - 1] >= 0;  // t - p >= 0
  Fq28<C> c = negv ? plus : (big ? minus : t);
  // repack 28 -> 32
#pragma unroll
  for (int j = 0; j < C::N; j++) {
    int bit = 32 * j;
    int lo = bit / 28, sh = bit % 28;
    uint64_t x = (uint64_t)(uint32_t)c.v[lo] >> sh;
    int got = 28 - sh;
    if (lo + 1 < C::L) x |= (uint64_t)(uint32_t)c.v[lo + 1] << got;
    if (lo + 2 < C::L && got + 28 < 32) x |= (uint64_t)(uint32_t)c.v[lo + 2] << (got + 28);
    w[j] = (uint32_t)x;
  }
}

}  // namespace gs
