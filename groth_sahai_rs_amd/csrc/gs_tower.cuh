// Extension tower Fp2 / Fp6 / Fp12 for pairing-friendly curves with u^2 = -1,
// v^3 = xi, w^2 = v (BLS12-381: xi = 1+u; BN254: xi = 9+u).
//
// Replaces arkworks' Fp2/Fp6/Fp12 models (ark-ff ^0.5, external to the
// reference; reached from ComT ops src/data_structures.rs:399-466 and from
// E::pairing / E::multi_pairing :484-502).  Element order in memory matches
// arkworks: Fp12{c0,c1: Fp6}, Fp6{c0,c1,c2: Fp2}, Fp2{c0,c1: Fp}.
//
// Fp2 values travel in registers (by value); Fp6/Fp12 operations are
// out-of-line and work on memory operands (an Fp12 is 144 dwords -- it cannot
// live in VGPRs next to anything else), which also keeps the hot loops within
// the instruction cache.
#pragma once
#include "gs_field.cuh"

namespace gs {

template <class C> using Fq = Fe<FqM<C>>;
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
template <class C> GS_HD bool is_zero(const Fp2<C>& a) { return is_zero(a.c0) && is_zero(a.c1); }
template <class C> GS_HD bool eq(const Fp2<C>& a, const Fp2<C>& b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
template <class C> GS_HD Fp2<C> select(bool c, const Fp2<C>& a, const Fp2<C>& b) {
  return {select(c, a.c0, b.c0), select(c, a.c1, b.c1)};
}
template <class C> GS_HD Fp2<C> mul(const Fp2<C>& a, const Fp2<C>& b) {
  Fq<C> v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
  Fq<C> s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
  return {sub(v0, v1), sub(sub(s, v0), v1)};
}
template <class C> GS_HD Fp2<C> sqr(const Fp2<C>& a) {
  Fq<C> t = mul(a.c0, a.c1);
  return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(t)};
}
template <class C> GS_HD Fp2<C> mul_fp(const Fp2<C>& a, const Fq<C>& k) { return {mul(a.c0, k), mul(a.c1, k)}; }
template <class C> GS_HD Fp2<C> mul_xi(const Fp2<C>& a) {
  if (C::XI_A == 1) return {sub(a.c0, a.c1), add(a.c0, a.c1)};
  // (A + u)(a0 + a1 u) = (A a0 - a1) + (A a1 + a0) u
  return {sub(mul_small(a.c0, C::XI_A), a.c1), add(mul_small(a.c1, C::XI_A), a.c0)};
}
template <class C> GS_HD Fp2<C> inv(const Fp2<C>& a) {
  Fq<C> n = inv(add(sqr(a.c0), sqr(a.c1)));
  return {mul(a.c0, n), neg(mul(a.c1, n))};
}
template <class C> GS_HD Fp2<C> mul_small(const Fp2<C>& a, int k) { return {mul_small(a.c0, k), mul_small(a.c1, k)}; }

template <class F> GS_HD F zero_of();
template <class F> GS_HD F one_of();
#define GS_ZERO_ONE(CURVE)                                                              \
  template <> GS_HD Fq<CURVE> zero_of<Fq<CURVE>>() { return fzero<FqM<CURVE>>(); }      \
  template <> GS_HD Fq<CURVE> one_of<Fq<CURVE>>() { return fone<FqM<CURVE>>(); }        \
  template <> GS_HD Fp2<CURVE> zero_of<Fp2<CURVE>>() {                                  \
    return {fzero<FqM<CURVE>>(), fzero<FqM<CURVE>>()};                                  \
  }                                                                                     \
  template <> GS_HD Fp2<CURVE> one_of<Fp2<CURVE>>() { return {fone<FqM<CURVE>>(), fzero<FqM<CURVE>>()}; }

// halve: (a + (a odd ? p : 0)) >> 1
template <class M> GS_HD Fe<M> half(const Fe<M>& a) {
  constexpr int N = M::N;
  uint32_t mask = 0u - (a.v[0] & 1u);
  uint32_t t[N];
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    uint64_t x = (uint64_t)a.v[j] + (M::mod(j) & mask) + c;
    t[j] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
  Fe<M> r;
#pragma unroll
  for (int j = 0; j < N - 1; j++) r.v[j] = (t[j] >> 1) | (t[j + 1] << 31);
  r.v[N - 1] = (t[N - 1] >> 1) | (c << 31);
  return r;
}
template <class C> GS_HD Fp2<C> half(const Fp2<C>& a) { return {half(a.c0), half(a.c1)}; }

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
// r = a * v   (v^3 = xi)
template <class C> GS_HD void f6_mul_v(Fp6<C>& r, const Fp6<C>& a) {
  Fp2<C> t = mul_xi(a.c2);
  r.c2 = a.c1;
  r.c1 = a.c0;
  r.c0 = t;
}
// Karatsuba, 6 Fp2 multiplications.  r may alias a or b.
template <class C> GS_HD_NOINLINE void f6_mul(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {
  Fp2<C> v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1), v2 = mul(a.c2, b.c2);
  Fp2<C> t0 = sub(sub(mul(add(a.c1, a.c2), add(b.c1, b.c2)), v1), v2);
  Fp2<C> t1 = sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1);
  Fp2<C> t2 = sub(sub(mul(add(a.c0, a.c2), add(b.c0, b.c2)), v0), v2);
  r.c0 = add(v0, mul_xi(t0));
  r.c1 = add(t1, mul_xi(v2));
  r.c2 = add(t2, v1);
}
// a * (b0 + b1 v): 5 Fp2 multiplications
template <class C> GS_HD_NOINLINE void f6_mul_by_01(Fp6<C>& r, const Fp6<C>& a, const Fp2<C>& b0, const Fp2<C>& b1) {
  Fp2<C> v0 = mul(a.c0, b0), v1 = mul(a.c1, b1);
  Fp2<C> t0 = sub(mul(add(a.c1, a.c2), b1), v1);                    // a2*b1
  Fp2<C> t1 = sub(sub(mul(add(a.c0, a.c1), add(b0, b1)), v0), v1);  // a0 b1 + a1 b0
  Fp2<C> t2 = sub(mul(add(a.c0, a.c2), b0), v0);                    // a2*b0
  r.c0 = add(v0, mul_xi(t0));
  r.c1 = t1;
  r.c2 = add(t2, v1);
}
// a * (b1 v): 3 Fp2 multiplications
template <class C> GS_HD void f6_mul_by_1(Fp6<C>& r, const Fp6<C>& a, const Fp2<C>& b1) {
  Fp2<C> t0 = mul_xi(mul(a.c2, b1));
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
  Fp2<C> t0 = sub(sqr(a.c0), mul_xi(mul(a.c1, a.c2)));
  Fp2<C> t1 = sub(mul_xi(sqr(a.c2)), mul(a.c0, a.c1));
  Fp2<C> t2 = sub(sqr(a.c1), mul(a.c0, a.c2));
  Fp2<C> n = add(mul(a.c0, t0), mul_xi(add(mul(a.c2, t1), mul(a.c1, t2))));
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
template <class C> GS_HD bool f12_eq(const Fp12<C>& a, const Fp12<C>& b) {
  return eq(a.c0.c0, b.c0.c0) && eq(a.c0.c1, b.c0.c1) && eq(a.c0.c2, b.c0.c2) && eq(a.c1.c0, b.c1.c0) &&
         eq(a.c1.c1, b.c1.c1) && eq(a.c1.c2, b.c1.c2);
}
template <class C> GS_HD bool f12_is_one(const Fp12<C>& a) {
  return eq(a.c0.c0, one_of<Fp2<C>>()) && is_zero(a.c0.c1) && is_zero(a.c0.c2) && is_zero(a.c1.c0) &&
         is_zero(a.c1.c1) && is_zero(a.c1.c2);
}
template <class C> GS_HD_NOINLINE void f12_mul(Fp12<C>& r, const Fp12<C>& a, const Fp12<C>& b) {
  Fp6<C> t0, t1, sa, sb, m;
  f6_mul(t0, a.c0, b.c0);
  f6_mul(t1, a.c1, b.c1);
  f6_add(sa, a.c0, a.c1);
  f6_add(sb, b.c0, b.c1);
  f6_mul(m, sa, sb);
  f6_sub(m, m, t0);
  f6_sub(r.c1, m, t1);
  f6_mul_v(t1, t1);
  f6_add(r.c0, t0, t1);
}
// complex squaring: 2 Fp6 multiplications
template <class C> GS_HD_NOINLINE void f12_sqr(Fp12<C>& r, const Fp12<C>& a) {
  Fp6<C> v0, s0, s1, t;
  f6_mul(v0, a.c0, a.c1);
  f6_add(s0, a.c0, a.c1);
  f6_mul_v(t, a.c1);
  f6_add(s1, a.c0, t);
  f6_mul(s0, s0, s1);  // (a0+a1)(a0+v a1) = a0^2 + v a1^2 + (1+v) a0 a1
  f6_sub(s0, s0, v0);
  f6_mul_v(t, v0);
  f6_sub(r.c0, s0, t);
  f6_add(r.c1, v0, v0);
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
  f6_inv(t1, t0);
  f6_mul(r.c0, a.c0, t1);
  f6_mul(t0, a.c1, t1);
  f6_neg(r.c1, t0);
}

template <class C> GS_HD Fp2<C> frob_coeff(int j, int k) {
  Fp2<C> r;
#pragma unroll
  for (int i = 0; i < C::N; i++) {
    r.c0.v[i] = (j == 1) ? C::FROB1[k][0][i] : (j == 2) ? C::FROB2[k][0][i] : C::FROB3[k][0][i];
    r.c1.v[i] = (j == 1) ? C::FROB1[k][1][i] : (j == 2) ? C::FROB2[k][1][i] : C::FROB3[k][1][i];
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

// f *= (l0 + l1 v) + (l4 v) w      -- line shape of an M-type twist (BLS12-381)
template <class C>
GS_HD_NOINLINE void f12_mul_by_014(Fp12<C>& f, const Fp2<C>& l0, const Fp2<C>& l1, const Fp2<C>& l4) {
  Fp6<C> aa, bb, s, t;
  f6_mul_by_01(aa, f.c0, l0, l1);
  f6_mul_by_1(bb, f.c1, l4);
  f6_add(s, f.c0, f.c1);
  f6_mul_by_01(t, s, l0, add(l1, l4));
  f6_sub(t, t, aa);
  f6_sub(f.c1, t, bb);
  f6_mul_v(bb, bb);
  f6_add(f.c0, aa, bb);
}
// f *= l0 + (l3 + l4 v) w          -- line shape of a D-type twist (BN254)
template <class C>
GS_HD_NOINLINE void f12_mul_by_034(Fp12<C>& f, const Fp2<C>& l0, const Fp2<C>& l3, const Fp2<C>& l4) {
  Fp6<C> aa, bb, s, t;
  f6_mul_fp2(aa, f.c0, l0);
  f6_mul_by_01(bb, f.c1, l3, l4);
  f6_add(s, f.c0, f.c1);
  f6_mul_by_01(t, s, add(l0, l3), l4);
  f6_sub(t, t, aa);
  f6_sub(f.c1, t, bb);
  f6_mul_v(bb, bb);
  f6_add(f.c0, aa, bb);
}

// Granger-Scott squaring for elements of the cyclotomic subgroup (after the
// easy part of the final exponentiation): 6 Fp2 squarings-worth of work.
template <class C> GS_HD void fp4_sqr(Fp2<C>& o0, Fp2<C>& o1, const Fp2<C>& a, const Fp2<C>& b) {
  // (a + b t)^2 with t^2 = xi:  o0 = a^2 + xi b^2, o1 = 2ab
  Fp2<C> ab = mul(a, b);
  o0 = sub(sub(mul(add(a, b), add(a, mul_xi(b))), ab), mul_xi(ab));
  o1 = dbl(ab);
}
template <class C> GS_HD_NOINLINE void f12_cyclo_sqr(Fp12<C>& r, const Fp12<C>& f) {
  Fp2<C> t0, t1, t2, t3, t4, t5;
  fp4_sqr(t0, t1, f.c0.c0, f.c1.c1);
  fp4_sqr(t2, t3, f.c1.c0, f.c0.c2);
  fp4_sqr(t4, t5, f.c0.c1, f.c1.c2);
  // z0 = 3 t0 - 2 z0 ; z1 = 3 t1 + 2 z1
  Fp2<C> z;
  z = sub(t0, f.c0.c0);
  r.c0.c0 = add(dbl(z), t0);
  z = add(t1, f.c1.c1);
  r.c1.c1 = add(dbl(z), t1);
  // z2 = 3 xi t5 + 2 z2 ; z3 = 3 t4 - 2 z3
  Fp2<C> x5 = mul_xi(t5);
  z = add(x5, f.c1.c0);
  r.c1.c0 = add(dbl(z), x5);
  z = sub(t4, f.c0.c2);
  r.c0.c2 = add(dbl(z), t4);
  // z4 = 3 t2 - 2 z4 ; z5 = 3 t3 + 2 z5
  z = sub(t2, f.c0.c1);
  r.c0.c1 = add(dbl(z), t2);
  z = add(t3, f.c1.c2);
  r.c1.c2 = add(dbl(z), t3);
}

}  // namespace gs
