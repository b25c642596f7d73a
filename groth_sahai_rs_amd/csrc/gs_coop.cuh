// Three-lane cooperative exponentiation by the curve parameter for the final
// exponentiation (E::pairing / E::multi_pairing tail, src/data_structures.rs:484-502).
//
// One lane per final exponentiation is a ~8 k-multiplication dependent chain; a
// batch of 2^12 equations has only 2^14 of them, a quarter of the SIMDs' lanes.
// The five (BLS12) / three (BN) f^|x| runs are 87 % of that chain, and Granger-Scott
// squaring acts on the three Fp4 "pairs" of an Fp12 element independently:
//
//   Fp4 = Fp2[s]/(s^2 - xi),  s = w^3;   Fp12 = Fp4[w]/(w^3 - s):
//   f = A0 + A1 w + A2 w^2,  A0 = (c0.c0, c1.c1), A1 = (c1.c0, c0.c2), A2 = (c0.c1, c1.c2)
//
// so lane j of a 3-lane group owns A_j.  A squaring is one local Fp4 squaring plus a
// swap of the results between lanes 1 and 2 ((A1 w)^2 lands on w^2, (A2 w^2)^2 on
// s w); a multiplication by the (replicated) base is three Fp4 products per lane
// after an all-gather of the accumulator: C_j = sum_i [i > j ? s : 1] A_i B_((j-i) mod 3).
// All lanes run the same instruction stream (operands are picked with selects).
// Everything outside the f^|x| runs stays replicated on the three lanes in the
// ordinary representation, which costs no latency and needs no exchange.
//
// The exchange is a policy: wave shuffles on the device (CoopWave), a barrier-
// synchronised mailbox in the CPU twin (tests/twin/host_twin.cpp).
#pragma once
#include "gs_tower.cuh"

namespace gs {

template <class C> struct Fp4 {
  Fp2<C> a, b;  // a + b s
};

template <class C> GS_HD Fp4<C> select(bool c, const Fp4<C>& x, const Fp4<C>& y) {
  return {select(c, x.a, y.a), select(c, x.b, y.b)};
}
// pair k (lane-varying) of a replicated Fp12
template <class C> GS_HD Fp4<C> fp4_of(const Fp12<C>& f, int k) {
  Fp4<C> p0 = {f.c0.c0, f.c1.c1}, p1 = {f.c1.c0, f.c0.c2}, p2 = {f.c0.c1, f.c1.c2};
  return select(k == 0, p0, select(k == 1, p1, p2));
}
// s * (a + b s) = xi b + a s ; input N, output N
template <class C> GS_HD Fp4<C> fp4_mul_s(const Fp4<C>& x) { return {norm(mul_xi(x.b)), x.a}; }
// inputs N, output: a with A <= 2 (lazy), b N -- callers sum three of these and normalise
template <class C> GS_HD Fp4<C> fp4_mul_lazy(const Fp4<C>& x, const Fp4<C>& y) {
  Fp2<C> v0 = mul(x.a, y.a), v1 = mul(x.b, y.b);
  Fp2<C> s = mul_l2(add(x.a, x.b), add(y.a, y.b));
  return {add(v0, norm(mul_xi(v1))), norm(sub(sub(s, v0), v1))};
}

// C_j of A * B given the gathered accumulator A[0..2] and the lane's pre-arranged base
// Bs[i] = [i > j ? s : 1] * B_((j - i) mod 3)
template <class C> GS_HD Fp4<C> c12_mul_lane(const Fp4<C> A[3], const Fp4<C> Bs[3]) {
  Fp4<C> t0 = fp4_mul_lazy(A[0], Bs[0]), t1 = fp4_mul_lazy(A[1], Bs[1]), t2 = fp4_mul_lazy(A[2], Bs[2]);
  return {norm(add(add(t0.a, t1.a), t2.a)), norm(add(add(t0.b, t1.b), t2.b))};
}
template <class C> GS_HD void c12_base_lane(Fp4<C> Bs[3], const Fp12<C>& f, int j) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int k = j - i + (j < i ? 3 : 0);
    Fp4<C> b = fp4_of(f, k);
    Bs[i] = select(i > j, fp4_mul_s(b), b);
  }
}
// Granger-Scott squaring, lane part 2: own pair z, received squares r (from lane j itself for
// j = 0, from the other of lanes 1/2 otherwise).  f12_cyclo_sqr's formulas, pair by pair:
//   j = 0, 2:  a' = 3 r.a - 2 z.a,      b' = 3 r.b + 2 z.b
//   j = 1   :  a' = 3 xi r.b + 2 z.a,   b' = 3 r.a - 2 z.b
template <class C> GS_HD Fp4<C> c12_sqr_finish(const Fp4<C>& z, const Fp4<C>& r, int j) {
  bool mid = (j == 1);
  Fp2<C> xr = norm(mul_xi(r.b));
  Fp2<C> ua = select(mid, xr, r.a), ub = select(mid, r.a, r.b);
  Fp2<C> za = select(mid, z.a, neg(z.a)), zb = select(mid, neg(z.b), z.b);
  Fp2<C> t = add(ua, za);
  Fp4<C> o;
  o.a = norm(add(dbl(t), ua));
  t = add(ub, zb);
  o.b = norm(add(dbl(t), ub));
  return o;
}
template <class C> GS_HD Fp4<C> fp4_vreduce(const Fp4<C>& x) {
  return {{vreduce(x.a.c0), vreduce(x.a.c1)}, {vreduce(x.b.c0), vreduce(x.b.c1)}};
}
template <class C> GS_HD void f12_from_pairs(Fp12<C>& f, const Fp4<C> A[3]) {
  f.c0.c0 = A[0].a;
  f.c1.c1 = A[0].b;
  f.c1.c0 = A[1].a;
  f.c0.c2 = A[1].b;
  f.c0.c1 = A[2].a;
  f.c1.c2 = A[2].b;
}

// One squaring / one multiplication by the base of the group's accumulator.  Out of line unless GS_FE_INLINE (with the
// multiplier an asm statement rather than a call, the squaring step is better off inside the loop: `acc` stays in
// registers from step to step).
#if defined(GS_FE_INLINE)
#define GS_C12 GS_HD
#else
#define GS_C12 GS_HD_NOINLINE
#endif
template <class C, class X> GS_C12 void c12_sqr_step(Fp4<C>& acc, int j, X& xch) {
  Fp4<C> t;
  fp4_sqr(t.a, t.b, acc.a, acc.b);
  Fp4<C> got = xch.swap12(t, j);
  acc = c12_sqr_finish(acc, got, j);
}
template <class C, class X> GS_HD_NOINLINE void c12_mul_step(Fp4<C>& acc, const Fp4<C>* Bs, int j, X& xch) {
  Fp4<C> A[3];
  xch.gather(A, acc, j);
  acc = c12_mul_lane(A, Bs);
}
template <class C> GS_C12 void c12_vreduce_step(Fp4<C>& acc) { acc = fp4_vreduce(acc); }

// r = f^x on a 3-lane group; f is replicated on the lanes, so is r.  X provides
//   Fp4 swap12(const Fp4& mine, int j)          value of lane (j == 0 ? 0 : 3 - j)
//   void gather(Fp4 A[3], const Fp4& mine, int j)
template <class C, class X> GS_HD_NOINLINE void f12_exp_by_x_coop(Fp12<C>& r, const Fp12<C>& f, int j, X& xch) {
  Fp4<C> Bs[3];
  c12_base_lane(Bs, f, j);
  Fp4<C> acc = fp4_of(f, j);
  int top = 63;
  while (!((C::X_ABS >> top) & 1)) top--;
  int since = 0;
  for (int i = top - 1; i >= 0; i--) {
    c12_sqr_step(acc, j, xch);
    if ((C::X_ABS >> i) & 1) {
      c12_mul_step(acc, Bs, j, xch);
      since = 0;
    } else if (++since == 3) {  // same value-growth discipline as f12_exp_by_x
      c12_vreduce_step(acc);
      since = 0;
    }
  }
  Fp4<C> A[3];
  xch.gather(A, acc, j);
  f12_from_pairs(r, A);
  if (C::X_NEG) f12_conj(r, r);
}

template <class C, class X> struct ExpXCoop {
  int j;
  X* x;
  GS_HD void operator()(Fp12<C>& r, const Fp12<C>& f) const { f12_exp_by_x_coop(r, f, j, *x); }
};

#if defined(__HIPCC__)
// exchange between the lanes g0, g0+1, g0+2 of one wave
struct CoopWave {
  int g0;
  template <class C> __device__ __forceinline__ Fq28<C> shfl(const Fq28<C>& x, int src) const {
    Fq28<C> r;
#pragma unroll
    for (int i = 0; i < C::L; i++) r.v[i] = __shfl(x.v[i], src, 64);
    return r;
  }
  template <class C> __device__ __forceinline__ Fp4<C> shfl4(const Fp4<C>& x, int src) const {
    return {{shfl(x.a.c0, src), shfl(x.a.c1, src)}, {shfl(x.b.c0, src), shfl(x.b.c1, src)}};
  }
  template <class C> __device__ __forceinline__ Fp4<C> swap12(const Fp4<C>& mine, int j) const {
    return shfl4(mine, g0 + (j == 0 ? 0 : 3 - j));
  }
  template <class C> __device__ __forceinline__ void gather(Fp4<C> A[3], const Fp4<C>& mine, int) const {
#pragma unroll
    for (int i = 0; i < 3; i++) A[i] = shfl4(mine, g0 + i);
  }
};
#endif

}  // namespace gs
