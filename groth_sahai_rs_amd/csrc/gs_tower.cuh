// Extension tower Fp2 / Fp6 / Fp12 for pairing-friendly curves with u^2 = -1,
// v^3 = xi, w^2 = v (BLS12-381: xi = 1+u; BN254: xi = 9+u).
//
// Replaces arkworks' Fp2/Fp6/Fp12 models (ark-ff ^0.5, external to the
// reference; reached from ComT ops src/data_structures.rs:399-466 and from
// E::pairing / E::multi_pairing :484-502).  Coefficient order matches arkworks:
// Fp12{c0,c1: Fp6}, Fp6{c0,c1,c2: Fp2}, Fp2{c0,c1: Fp}.
//
// Base field = Fq28 (gs_fq28.cuh): lazily reduced radix-2^28 limbs.  Limb-growth
// discipline ("A" = max |limb| / 2^28):
//   * every multiplication returns A ~ 1 ("N");  add/sub/neg/dbl/mul_xi are lazy
//     (A adds up) and must stay below A = 7.9 (int32);
//   * Fq mul needs A*B <= 8;  Fp2 mul/sqr need N inputs (Karatsuba sums double A),
//     mul_l2/sqr_l2 accept A <= 2 inputs and normalise their internal sums;
//   * Fp6 / Fp12 functions take N inputs and return N outputs.
// The CPU twin built with -DGS_FQ28_CHECK asserts all of this at run time.
//
// Fp2 values travel in registers (by value); Fp6/Fp12 operations are
// out-of-line and work on memory operands (an Fp12 is 168 dwords).
#pragma once
#include "gs_fq28.cuh"

// Fp6 products inline into their Fp12 callers when GS_F6_INLINE is set (keeps the Fp6
// temporaries in the register file instead of handing them over through memory)
#if defined(GS_F6_INLINE)
#define GS_F6 GS_HD
#else
#define GS_F6 GS_HD_NOINLINE
#endif
// experiment: inline the Fp12 squaring / sparse product and the Miller steps into the loop
#if (defined(GS_MILLER_INLINE) || defined(GS_FE_INLINE)) && !defined(GS_LONGBR_OK) && defined(__HIPCC__)
#error "GS_MILLER_INLINE / GS_FE_INLINE need -mllvm -amdgpu-long-branch-factor=0 (csrc/Makefile probes it and passes -DGS_LONGBR_OK): without it the reserved long-branch register aliases the return address and the kernels never return"
#endif
#if defined(GS_MILLER_INLINE)
#define GS_ML GS_HD
#else
#define GS_ML GS_HD_NOINLINE
#endif

namespace gs {

template <class C> using Fq = Fq28<C>;
template <class C> using Fr = Fe<FrM<C>>;

// ------------------------------------------------------------------ Fp2 ----
template <class C> struct Fp2 {
  Fq<C> c0, c1;
};

template <class C> GS_HD Fp2<C> add(const Fp2<C>& a, const Fp2<C>& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
template <class C> GS_HD Fp2<C> sub(const Fp2<C>& a, const Fp2<C>& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
template <class C> GS_HD Fp2<C> neg(const Fp2<C>& a) { return {neg(a.c0), neg(a.c1)}; }
template <class C> GS_HD Fp2<C> dbl(const Fp2<C>& a) { return {dbl(a.c0), dbl(a.c1)}; }
template <class C> GS_HD Fp2<C> conj(const Fp2<C>& a) { return {a.c0, neg(a.c1)}; }
template <class C> GS_HD Fp2<C> norm(const Fp2<C>& a) { return {norm(a.c0), norm(a.c1)}; }
template <class C> GS_HD Fp2<C> mul_small(const Fp2<C>& a, int k) { return {mul_small(a.c0, k), mul_small(a.c1, k)}; }
template <class C> GS_HD bool is_zero(const Fp2<C>& a) { return is_zero(a.c0) && is_zero(a.c1); }
template <class C> GS_HD bool is_zero_limbs(const Fp2<C>& a) { return is_zero_limbs(a.c0) && is_zero_limbs(a.c1); }
template <class C> GS_HD bool eq(const Fp2<C>& a, const Fp2<C>& b) { return is_zero(sub(a, b)); }
template <class C> GS_HD Fp2<C> select(bool c, const Fp2<C>& a, const Fp2<C>& b) {
  return {select(c, a.c0, b.c0), select(c, a.c1, b.c1)};
}
#if !defined(GS_FP2_KARATSUBA)
// Schoolbook with lazy reduction as ONE multiplier kernel (fp2mul28, gs_fq28.cuh): inputs with A_a * A_b <= 4 -- N
// inputs and sums of two N values alike -- both outputs N.
template <class C> GS_HD Fp2<C> mul(const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> r;
  fp2mul28<C>(r.c0, r.c1, a.c0, a.c1, b.c0, b.c1);
  return r;
}
template <class C> GS_HD Fp2<C> mul_l2(const Fp2<C>& a, const Fp2<C>& b) { return mul(a, b); }
#else
// Karatsuba; inputs N, output N
template <class C> GS_HD Fp2<C> mul(const Fp2<C>& a, const Fp2<C>& b) {
  Fq<C> v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
  Fq<C> s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
  return {sub(v0, v1), norm(sub(sub(s, v0), v1))};
}
// inputs with A <= 2 (e.g. one sum of two N values)
template <class C> GS_HD Fp2<C> mul_l2(const Fp2<C>& a, const Fp2<C>& b) {
  Fq<C> v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
  // one carry round is enough: (A <= 4) x (A ~ 1) stays inside the multiplier's A*B <= 8 contract
  Fq<C> s = mul(add(a.c0, a.c1), norm(add(b.c0, b.c1)));
  return {sub(v0, v1), norm(sub(sub(s, v0), v1))};
}
#endif
#if !defined(GS_FP2_KARATSUBA)
// one multiplier kernel (fp2sqr28): inputs with A <= 2, both outputs N
template <class C> GS_HD Fp2<C> sqr(const Fp2<C>& a) {
  Fp2<C> r;
  fp2sqr28<C>(r.c0, r.c1, a.c0, a.c1);
  return r;
}
template <class C> GS_HD Fp2<C> sqr_l2(const Fp2<C>& a) { return sqr(a); }
#else
template <class C> GS_HD Fp2<C> sqr(const Fp2<C>& a) {
  Fq<C> t = mul(a.c0, a.c1);
  return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), norm(dbl(t))};
}
template <class C> GS_HD Fp2<C> sqr_l2(const Fp2<C>& a) {
  Fq<C> t = mul(a.c0, a.c1);
  return {mul(add(a.c0, a.c1), norm(sub(a.c0, a.c1))), norm(dbl(t))};
}
#endif
// a0 b0 + a1 b1 + a2 b2 as ONE multiplier kernel (fp2dot3_28): operands N (sum of A_a A_b <= 4), output N
template <class C>
GS_HD Fp2<C> dot3(const Fp2<C>& a0, const Fp2<C>& b0, const Fp2<C>& a1, const Fp2<C>& b1, const Fp2<C>& a2,
                  const Fp2<C>& b2) {
  Fp2<C> r;
  fp2dot3_28<C>(r.c0, r.c1, a0.c0, a0.c1, b0.c0, b0.c1, a1.c0, a1.c1, b1.c0, b1.c1, a2.c0, a2.c1, b2.c0, b2.c1);
  return r;
}
template <class C> GS_HD Fp2<C> mul_fp(const Fp2<C>& a, const Fq<C>& k) { return {mul(a.c0, k), mul(a.c1, k)}; }
// lazy: A_out = (XI_A + 1) * A_in
template <class C> GS_HD Fp2<C> mul_xi(const Fp2<C>& a) {
  if (C::XI_A == 1) return {sub(a.c0, a.c1), add(a.c0, a.c1)};
  // (9 + u)(a0 + a1 u) = (9 a0 - a1) + (9 a1 + a0) u ; 8x is normalised on the way (A_in <= 1.9)
  Fq<C> e0 = norm(mul_small(a.c0, 4)), e1 = norm(mul_small(a.c1, 4));
  e0 = norm(dbl(e0));
  e1 = norm(dbl(e1));
  return {sub(add(e0, a.c0), a.c1), add(add(e1, a.c1), a.c0)};
}
template <class C> GS_HD Fp2<C> inv(const Fp2<C>& a) {
  Fq<C> n = inv(add(sqr(a.c0), sqr(a.c1)));
  return {mul(a.c0, n), neg(mul(a.c1, n))};
}

// generic helpers for code templated on F = Fq or Fp2
template <class C> GS_HD Fq<C> mul_l2(const Fq<C>& a, const Fq<C>& b) { return mul(a, b); }  // 2*2 <= 8
template <class C> GS_HD Fq<C> sqr_l2(const Fq<C>& a) { return sqr(a); }

template <class F> GS_HD F zero_of();
template <class F> GS_HD F one_of();
#define GS_ZERO_ONE(CURVE)                                                                     \
  template <> GS_HD Fq<CURVE> zero_of<Fq<CURVE>>() { return fq_zero<CURVE>(); }                \
  template <> GS_HD Fq<CURVE> one_of<Fq<CURVE>>() { return fq_one<CURVE>(); }                  \
  template <> GS_HD Fp2<CURVE> zero_of<Fp2<CURVE>>() { return {fq_zero<CURVE>(), fq_zero<CURVE>()}; } \
  template <> GS_HD Fp2<CURVE> one_of<Fp2<CURVE>>() { return {fq_one<CURVE>(), fq_zero<CURVE>()}; }

// ------------------------------------------------------------------ Fp6 ----
template <class C> struct Fp6 {
  Fp2<C> c0, c1, c2;
};

template <class C> GS_HD void f6_add(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {
  r.c0 = add(a.c0, b.c0);
  r.c1 = add(a.c1, b.c1);
  r.c2 = add(a.c2, b.c2);
}
template <class C> GS_HD void f6_sub(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {
  r.c0 = sub(a.c0, b.c0);
  r.c1 = sub(a.c1, b.c1);
  r.c2 = sub(a.c2, b.c2);
}
template <class C> GS_HD void f6_neg(Fp6<C>& r, const Fp6<C>& a) {
  r.c0 = neg(a.c0);
  r.c1 = neg(a.c1);
  r.c2 = neg(a.c2);
}
template <class C> GS_HD void f6_norm(Fp6<C>& r, const Fp6<C>& a) {
  r.c0 = norm(a.c0);
  r.c1 = norm(a.c1);
  r.c2 = norm(a.c2);
}
// r = norm(a + b), r = norm(a - b)
template <class C> GS_HD void f6_addn(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {
  r.c0 = norm(add(a.c0, b.c0));
  r.c1 = norm(add(a.c1, b.c1));
  r.c2 = norm(add(a.c2, b.c2));
}
// r = a * v   (v^3 = xi); N in -> N out
template <class C> GS_HD void f6_mul_v(Fp6<C>& r, const Fp6<C>& a) {
  Fp2<C> t = norm(mul_xi(a.c2));
  r.c2 = a.c1;
  r.c1 = a.c0;
  r.c0 = t;
}
// Karatsuba, 6 Fp2 multiplications.  r may alias a or b.  N in -> N out.
template <class C> GS_F6 void f6_mul(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {
  Fp2<C> v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1), v2 = mul(a.c2, b.c2);
  Fp2<C> t0 = sub(sub(mul_l2(add(a.c1, a.c2), add(b.c1, b.c2)), v1), v2);  // A = 3
  Fp2<C> t1 = sub(sub(mul_l2(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1);
  Fp2<C> t2 = sub(sub(mul_l2(add(a.c0, a.c2), add(b.c0, b.c2)), v0), v2);
  // xi = 1 + u: xi t0 has A <= 6 and the sum A <= 7, one carry round at the end is enough; xi = 9 + u needs A ~ 1 in
  r.c0 = norm(add(v0, mul_xi(C::XI_A == 1 ? t0 : norm(t0))));
  r.c1 = norm(add(t1, mul_xi(v2)));
  r.c2 = norm(add(t2, v1));
}
// a * (b0 + b1 v): 5 Fp2 multiplications
template <class C> GS_F6 void f6_mul_by_01(Fp6<C>& r, const Fp6<C>& a, const Fp2<C>& b0, const Fp2<C>& b1) {
  Fp2<C> v0 = mul(a.c0, b0), v1 = mul(a.c1, b1);
  Fp2<C> t0 = mul(a.c2, b1), t2 = mul(a.c2, b0);                           // only a0 b1 + a1 b0 gains from Karatsuba
  Fp2<C> t1 = sub(sub(mul_l2(add(a.c0, a.c1), add(b0, b1)), v0), v1);
  r.c0 = norm(add(v0, mul_xi(t0)));
  r.c1 = norm(t1);
  r.c2 = norm(add(t2, v1));
}
// a * (b1 v): 3 Fp2 multiplications
template <class C> GS_HD void f6_mul_by_1(Fp6<C>& r, const Fp6<C>& a, const Fp2<C>& b1) {
  Fp2<C> t0 = norm(mul_xi(mul(a.c2, b1)));
  Fp2<C> t1 = mul(a.c0, b1);
  Fp2<C> t2 = mul(a.c1, b1);
  r.c0 = t0;
  r.c1 = t1;
  r.c2 = t2;
}
template <class C> GS_HD void f6_mul_fp2(Fp6<C>& r, const Fp6<C>& a, const Fp2<C>& k) {
  r.c0 = mul(a.c0, k);
  r.c1 = mul(a.c1, k);
  r.c2 = mul(a.c2, k);
}
template <class C> GS_HD_NOINLINE void f6_inv(Fp6<C>& r, const Fp6<C>& a) {
  Fp2<C> t0 = norm(sub(sqr(a.c0), mul_xi(mul(a.c1, a.c2))));
  Fp2<C> t1 = norm(sub(norm(mul_xi(sqr(a.c2))), mul(a.c0, a.c1)));
  Fp2<C> t2 = norm(sub(sqr(a.c1), mul(a.c0, a.c2)));
  Fp2<C> s = norm(add(mul(a.c2, t1), mul(a.c1, t2)));
  Fp2<C> n = norm(add(mul(a.c0, t0), mul_xi(s)));
  Fp2<C> ni = inv(n);
  r.c0 = mul(t0, ni);
  r.c1 = mul(t1, ni);
  r.c2 = mul(t2, ni);
}

// ----------------------------------------------------------------- Fp12 ----
template <class C> struct Fp12 {
  Fp6<C> c0, c1;
};

template <class C> GS_HD void f12_one(Fp12<C>& r) {
  Fp2<C> z = zero_of<Fp2<C>>();
  r.c0.c0 = one_of<Fp2<C>>();
  r.c0.c1 = z;
  r.c0.c2 = z;
  r.c1.c0 = z;
  r.c1.c1 = z;
  r.c1.c2 = z;
}
// GS_FE_INLINE: the x-power loops of the final exponentiation take the Fp12 product and the cyclotomic squaring
// INLINE (the accumulator then stays in registers from one step to the next instead of crossing memory at every call)
template <class C> GS_HD void f12_mul_inl(Fp12<C>& r, const Fp12<C>& a, const Fp12<C>& b) {
  Fp6<C> t0, t1, sa, sb, m;
  f6_mul(t0, a.c0, b.c0);
  f6_mul(t1, a.c1, b.c1);
  f6_addn(sa, a.c0, a.c1);
  f6_addn(sb, b.c0, b.c1);
  f6_mul(m, sa, sb);
  f6_sub(m, m, t0);
  f6_sub(m, m, t1);
  f6_norm(r.c1, m);
  f6_mul_v(t1, t1);
  f6_addn(r.c0, t0, t1);
}
// (Round 4, measured and not kept: the out-of-line product as three calls of a GENERATED Fp6 product -- gen_pointops_asm.py
// f6_mul: six nested calls of the shared Fp2 product with its own data movement, `a` in/out in VGPRs, `b` in AGPRs; the
// emitted code is checked by tests/test_pointops_gen.py.  k_final 36.3 -> 37.1 ms: hipcc's marshalling of 18 blocks per
// call and the ~590 callee-saved registers an out-of-line function must save around subroutines that clobber v0..v190 and
// a0..a195 cost more than the spills they replace.  GS_TOWER_ASM_EMIT=1 in the generator's environment brings the
// subroutine back.)
template <class C> GS_HD_NOINLINE void f12_mul(Fp12<C>& r, const Fp12<C>& a, const Fp12<C>& b) { f12_mul_inl(r, a, b); }
// complex squaring: 2 Fp6 multiplications
template <class C> GS_ML void f12_sqr(Fp12<C>& r, const Fp12<C>& a) {
  Fp6<C> v0, s0, s1, t;
  f6_mul(v0, a.c0, a.c1);
  f6_addn(s0, a.c0, a.c1);
  f6_mul_v(t, a.c1);
  f6_addn(s1, a.c0, t);
  f6_mul(s0, s0, s1);  // (a0+a1)(a0+v a1) = a0^2 + v a1^2 + (1+v) a0 a1
  f6_mul_v(t, v0);
  f6_sub(s0, s0, v0);
  f6_sub(s0, s0, t);
  f6_norm(r.c0, s0);
  f6_addn(r.c1, v0, v0);
}
template <class C> GS_HD void f12_conj(Fp12<C>& r, const Fp12<C>& a) {
  r.c0 = a.c0;
  f6_neg(r.c1, a.c1);
}
template <class C> GS_HD_NOINLINE void f12_inv(Fp12<C>& r, const Fp12<C>& a) {
  Fp6<C> t0, t1;
  f6_mul(t0, a.c0, a.c0);
  f6_mul(t1, a.c1, a.c1);
  f6_mul_v(t1, t1);
  f6_sub(t0, t0, t1);
  f6_norm(t0, t0);
  f6_inv(t1, t0);
  f6_mul(r.c0, a.c0, t1);
  f6_mul(t0, a.c1, t1);
  f6_neg(r.c1, t0);
}

template <class C> GS_HD Fp2<C> frob_coeff(int j, int k) {
  Fp2<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) {
    r.c0.v[i] = (j == 1) ? C::FROB1_28[k][0][i] : (j == 2) ? C::FROB2_28[k][0][i] : C::FROB3_28[k][0][i];
    r.c1.v[i] = (j == 1) ? C::FROB1_28[k][1][i] : (j == 2) ? C::FROB2_28[k][1][i] : C::FROB3_28[k][1][i];
  }
  return r;
}
// a^(p^j), j in {1,2,3}: coefficient of w^k (k = 2*vpow + wpow) is conjugated j
// times and multiplied by xi^(k (p^j - 1)/6).
template <class C> GS_HD_NOINLINE void f12_frob(Fp12<C>& r, const Fp12<C>& a, int j) {
  bool cj = (j & 1);
  Fp2<C> x;
  x = cj ? conj(a.c0.c0) : a.c0.c0;
  r.c0.c0 = x;
  x = cj ? conj(a.c0.c1) : a.c0.c1;
  r.c0.c1 = mul(x, frob_coeff<C>(j, 2));
  x = cj ? conj(a.c0.c2) : a.c0.c2;
  r.c0.c2 = mul(x, frob_coeff<C>(j, 4));
  x = cj ? conj(a.c1.c0) : a.c1.c0;
  r.c1.c0 = mul(x, frob_coeff<C>(j, 1));
  x = cj ? conj(a.c1.c1) : a.c1.c1;
  r.c1.c1 = mul(x, frob_coeff<C>(j, 3));
  x = cj ? conj(a.c1.c2) : a.c1.c2;
  r.c1.c2 = mul(x, frob_coeff<C>(j, 5));
}

// f *= (l0 + l1 v) + (l4 v) w      -- line shape of an M-type twist (BLS12-381); l* are N
template <class C>
GS_ML void f12_mul_by_014(Fp12<C>& f, const Fp2<C>& l0, const Fp2<C>& l1, const Fp2<C>& l4) {
#if !defined(GS_NO_SPARSE_DOT3)
  // schoolbook over the six coefficients: with A = a0 + a1 v + a2 v^2 = f.c0, B = f.c1,
  //   c0' = A (l0 + l1 v) + v B (l4 v),  c1' = B (l0 + l1 v) + A (l4 v)
  // every coefficient is a sum of three Fp2 products = one dot3 kernel (14 L^2 multiply-adds, output N): 84 L^2 against
  // the 78 L^2 of the Karatsuba form below, and none of its ~3 500 additions, carry rounds and copies around them.
  const Fp2<C> a0 = f.c0.c0, a1 = f.c0.c1, a2 = f.c0.c2, b0 = f.c1.c0, b1 = f.c1.c1, b2 = f.c1.c2;
  // Call order (round 3): the calls that take xi-multiples come first and next to each other, every xi-multiple is
  // dead after the third call, the results are written back at the end.  Live set: six old coefficients + the line +
  // the outputs so far + two xi-multiples + the multiplier's own registers = ~440 dwords; with all three xi-multiples
  // computed up front and the outputs stored as they came it was ~550 > 512, and every line product moved ~410 dwords
  // through the private segment (now ~310): k_miller.pairdpp 167.1 -> 158.5 ms at 2^16, same box.
  Fp2<C> n00, n01, n02, n10, n11, n12;
  {
    const Fp2<C> xa2 = norm(mul_xi(a2));
    {
      const Fp2<C> xb1 = norm(mul_xi(b1));
      n00 = dot3(a0, l0, xa2, l1, xb1, l4);
    }
    const Fp2<C> xb2 = norm(mul_xi(b2));
    n10 = dot3(b0, l0, xb2, l1, xa2, l4);
    n01 = dot3(a1, l0, a0, l1, xb2, l4);
  }
  n02 = dot3(a2, l0, a1, l1, b0, l4);
  n11 = dot3(b1, l0, b0, l1, a0, l4);
  n12 = dot3(b2, l0, b1, l1, a1, l4);
  f.c0.c0 = n00;
  f.c0.c1 = n01;
  f.c0.c2 = n02;
  f.c1.c0 = n10;
  f.c1.c1 = n11;
  f.c1.c2 = n12;
  return;
#endif
  Fp6<C> aa, bb, s, t;
  f6_mul_by_01(aa, f.c0, l0, l1);
  f6_mul_by_1(bb, f.c1, l4);
  f6_addn(s, f.c0, f.c1);
  f6_mul_by_01(t, s, l0, norm(add(l1, l4)));
  f6_sub(t, t, aa);
  f6_sub(t, t, bb);
  f6_norm(f.c1, t);
  f6_mul_v(bb, bb);
  f6_addn(f.c0, aa, bb);
}
// f *= l0 + (l3 + l4 v) w          -- line shape of a D-type twist (BN254)
template <class C>
GS_ML void f12_mul_by_034(Fp12<C>& f, const Fp2<C>& l0, const Fp2<C>& l3, const Fp2<C>& l4) {
#if !defined(GS_NO_SPARSE_DOT3)
  //   c0' = A l0 + v B (l3 + l4 v),  c1' = A (l3 + l4 v) + B l0
  const Fp2<C> a0 = f.c0.c0, a1 = f.c0.c1, a2 = f.c0.c2, b0 = f.c1.c0, b1 = f.c1.c1, b2 = f.c1.c2;
  // (same call order as f12_mul_by_014: xi-multiples first and short-lived, results written back at the end)
  Fp2<C> n00, n01, n02, n10, n11, n12;
  {
    const Fp2<C> xb2 = norm(mul_xi(b2));
    {
      const Fp2<C> xb1 = norm(mul_xi(b1));
      n00 = dot3(a0, l0, xb2, l3, xb1, l4);
    }
    n01 = dot3(a1, l0, b0, l3, xb2, l4);
  }
  {
    const Fp2<C> xa2 = norm(mul_xi(a2));
    n10 = dot3(b0, l0, a0, l3, xa2, l4);
  }
  n02 = dot3(a2, l0, b1, l3, b0, l4);
  n11 = dot3(b1, l0, a1, l3, a0, l4);
  n12 = dot3(b2, l0, a2, l3, a1, l4);
  f.c0.c0 = n00;
  f.c0.c1 = n01;
  f.c0.c2 = n02;
  f.c1.c0 = n10;
  f.c1.c1 = n11;
  f.c1.c2 = n12;
  return;
#endif
  Fp6<C> aa, bb, s, t;
  f6_mul_fp2(aa, f.c0, l0);
  f6_mul_by_01(bb, f.c1, l3, l4);
  f6_addn(s, f.c0, f.c1);
  f6_mul_by_01(t, s, norm(add(l0, l3)), l4);
  f6_sub(t, t, aa);
  f6_sub(t, t, bb);
  f6_norm(f.c1, t);
  f6_mul_v(bb, bb);
  f6_addn(f.c0, aa, bb);
}

// An evaluated Miller line: the three coefficients f12_mul_by_014 (M-type twist) / f12_mul_by_034 (D-type) take.
template <class C> struct ELine {
  Fp2<C> a, b, c;  // M: (l0 + l1 v) + (l4 v) w  as (l0, l1, l4);   D: l0 + (l3 + l4 v) w  as (l0, l3, l4)
};
template <class C> GS_HD void f12_mul_by_line(Fp12<C>& f, const ELine<C>& e) {
  if (C::TWIST_M)
    f12_mul_by_014(f, e.a, e.b, e.c);
  else
    f12_mul_by_034(f, e.a, e.b, e.c);
}
// f *= u * v for two evaluated lines.  The product of two lines costs 6 Fp2 multiplications and has five non-zero
// coefficients (c0 dense, c1 with two):
//   M: c0 = (aa' + xi cc', ab' + ba', bb'),  c1 = (0, ac' + ca', bc' + cb')
//   D: c0 = (aa' + xi cc', bb', bc' + cb'),  c1 = (ab' + ba', ac' + ca', 0)
// and multiplying f by it costs 6 + 5 + 6: 23 Fp2 multiplications for two lines instead of 2 x 13.
template <class C> GS_ML void f12_mul_by_lines2(Fp12<C>& f, const ELine<C>& u, const ELine<C>& v) {
  Fp2<C> aa = mul(u.a, v.a), bb = mul(u.b, v.b), cc = mul(u.c, v.c);
  Fp2<C> ab = norm(sub(sub(mul_l2(add(u.a, u.b), add(v.a, v.b)), aa), bb));
  Fp2<C> ac = norm(sub(sub(mul_l2(add(u.a, u.c), add(v.a, v.c)), aa), cc));
  Fp2<C> bc = norm(sub(sub(mul_l2(add(u.b, u.c), add(v.b, v.c)), bb), cc));
  Fp6<C> c0, s, t0, t1, sa, m;
  Fp2<C> y0, y1;
  c0.c0 = norm(add(aa, mul_xi(cc)));
  if (C::TWIST_M) {
    c0.c1 = ab;
    c0.c2 = norm(bb);
    y0 = ac;
    y1 = bc;
    s.c0 = c0.c0;
    s.c1 = norm(add(c0.c1, y0));
    s.c2 = norm(add(c0.c2, y1));
  } else {
    c0.c1 = norm(bb);
    c0.c2 = bc;
    y0 = ab;
    y1 = ac;
    s.c0 = norm(add(c0.c0, y0));
    s.c1 = norm(add(c0.c1, y1));
    s.c2 = c0.c2;
  }
  f6_mul(t0, f.c0, c0);
  f6_mul_by_01(t1, f.c1, y0, y1);       // D: f1 (y0 + y1 v)
  if (C::TWIST_M) f6_mul_v(t1, t1);     // M: f1 (y0 v + y1 v^2)
  f6_addn(sa, f.c0, f.c1);
  f6_mul(m, sa, s);
  f6_sub(m, m, t0);
  f6_sub(m, m, t1);
  f6_norm(f.c1, m);
  f6_mul_v(t1, t1);
  f6_addn(f.c0, t0, t1);
}

// Granger-Scott squaring for elements of the cyclotomic subgroup (after the
// easy part of the final exponentiation).
template <class C> GS_HD void fp4_sqr(Fp2<C>& o0, Fp2<C>& o1, const Fp2<C>& a, const Fp2<C>& b) {
  // (a + b t)^2 with t^2 = xi:  o0 = a^2 + xi b^2, o1 = 2ab ; outputs N
  // limb growth: a + b has A = 2 and meets a normalised factor (Fp2 product contract A_a A_b <= 4); the sum
  // m - ab - xi ab stays within 3 x 2^28 per limb before its one carry round
  Fp2<C> ab = mul(a, b);
  Fp2<C> m = mul(add(a, b), norm(add(a, mul_xi(b))));
  o0 = norm(sub(sub(m, ab), mul_xi(ab)));
  o1 = norm(dbl(ab));
}
template <class C> GS_HD void f12_cyclo_sqr_inl(Fp12<C>& r, const Fp12<C>& f) {
  Fp2<C> t0, t1, t2, t3, t4, t5;
  fp4_sqr(t0, t1, f.c0.c0, f.c1.c1);
  fp4_sqr(t2, t3, f.c1.c0, f.c0.c2);
  fp4_sqr(t4, t5, f.c0.c1, f.c1.c2);
  // z0 = 3 t0 - 2 z0 ; z1 = 3 t1 + 2 z1
  Fp2<C> z;
  z = sub(t0, f.c0.c0);
  r.c0.c0 = norm(add(dbl(z), t0));
  z = add(t1, f.c1.c1);
  r.c1.c1 = norm(add(dbl(z), t1));
  // z2 = 3 xi t5 + 2 z2 ; z3 = 3 t4 - 2 z3
  Fp2<C> x5 = norm(mul_xi(t5));
  z = add(x5, f.c1.c0);
  r.c1.c0 = norm(add(dbl(z), x5));
  z = sub(t4, f.c0.c2);
  r.c0.c2 = norm(add(dbl(z), t4));
  // z4 = 3 t2 - 2 z4 ; z5 = 3 t3 + 2 z5
  z = sub(t2, f.c0.c1);
  r.c0.c1 = norm(add(dbl(z), t2));
  z = add(t3, f.c1.c2);
  r.c1.c2 = norm(add(dbl(z), t3));
}
template <class C> GS_HD_NOINLINE void f12_cyclo_sqr(Fp12<C>& r, const Fp12<C>& f) { f12_cyclo_sqr_inl(r, f); }

// bring every coefficient's VALUE back to ~[-p, p] (see vreduce in gs_fq28.cuh)
template <class C> GS_HD void f12_vreduce_inl(Fp12<C>& f) {
#define GS_VR2(x) x.c0 = vreduce(x.c0), x.c1 = vreduce(x.c1)
  GS_VR2(f.c0.c0);
  GS_VR2(f.c0.c1);
  GS_VR2(f.c0.c2);
  GS_VR2(f.c1.c0);
  GS_VR2(f.c1.c1);
  GS_VR2(f.c1.c2);
#undef GS_VR2
}
template <class C> GS_HD_NOINLINE void f12_vreduce(Fp12<C>& f) { f12_vreduce_inl(f); }

// ---- boundary I/O (include/gs_amd.h layout: saturated Montgomery limbs) -------
template <class C> struct BFq {
  uint32_t w[C::N];
};
template <class C> GS_HD Fp2<C> fp2_from_boundary(const BFq<C>* p) {
  return {fq_from_boundary<C>(p[0].w), fq_from_boundary<C>(p[1].w)};
}
template <class C> GS_HD void fp2_to_boundary(BFq<C>* p, const Fp2<C>& a) {
  fq_to_boundary<C>(p[0].w, a.c0);
  fq_to_boundary<C>(p[1].w, a.c1);
}
template <class C> GS_HD_NOINLINE void f12_from_boundary(Fp12<C>& r, const BFq<C>* p) {
  r.c0.c0 = fp2_from_boundary<C>(p + 0);
  r.c0.c1 = fp2_from_boundary<C>(p + 2);
  r.c0.c2 = fp2_from_boundary<C>(p + 4);
  r.c1.c0 = fp2_from_boundary<C>(p + 6);
  r.c1.c1 = fp2_from_boundary<C>(p + 8);
  r.c1.c2 = fp2_from_boundary<C>(p + 10);
}
template <class C> GS_HD_NOINLINE void f12_to_boundary(BFq<C>* p, const Fp12<C>& a) {
  fp2_to_boundary<C>(p + 0, a.c0.c0);
  fp2_to_boundary<C>(p + 2, a.c0.c1);
  fp2_to_boundary<C>(p + 4, a.c0.c2);
  fp2_to_boundary<C>(p + 6, a.c1.c0);
  fp2_to_boundary<C>(p + 8, a.c1.c1);
  fp2_to_boundary<C>(p + 10, a.c1.c2);
}
// equality of two GT values through their canonical boundary forms
template <class C> GS_HD_NOINLINE bool f12_eq(const Fp12<C>& a, const Fp12<C>& b) {
  BFq<C> x[12], y[12];
  f12_to_boundary<C>(x, a);
  f12_to_boundary<C>(y, b);
  uint32_t o = 0;
  for (int i = 0; i < 12; i++)
    for (int j = 0; j < C::N; j++) o |= x[i].w[j] ^ y[i].w[j];
  return o == 0;
}
template <class C> GS_HD bool f12_is_one(const Fp12<C>& a) {
  Fp12<C> o;
  f12_one(o);
  return f12_eq(a, o);
}

}  // namespace gs
