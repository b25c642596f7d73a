// Short-Weierstrass (a = 0) group arithmetic over F = Fq (G1) or Fp2 (G2 twist).
//
// Replaces arkworks' short_weierstrass::{Affine,Projective} (ark-ec ^0.5,
// external to the reference): `into_group`, `*= scalar`, `into_affine`, `+`,
// `-` as used at src/data_structures.rs:187-188,197-198,208-209,337-341,382-386.
// Affine identity is the all-zero pair (0,0) (never on y^2 = x^3 + b, b != 0);
// Jacobian identity is Z = 0.  Every output leaves as the unique normalised
// affine point, so any correct addition chain is bit-exact with arkworks.
#pragma once
#include "gs_tower.cuh"

// experiment/option: inline the Jacobian formulas into the scalar-multiplication loops so that the
// running point stays in the register file (single call sites: the loops below are not unrolled)
#if defined(GS_JAC_INLINE)
#define GS_JAC GS_HD
#else
#define GS_JAC GS_HD_NOINLINE
#endif

namespace gs {

template <class F> struct Aff {
  F x, y;
};
template <class F> struct Jac {
  F x, y, z;
};

// identity flags are EXACT zeros (never lazily reduced representatives): see gs_fq28.cuh
template <class F> GS_HD bool aff_is_inf(const Aff<F>& p) { return is_zero_limbs(p.x) && is_zero_limbs(p.y); }
template <class F> GS_HD void jac_set_inf(Jac<F>& r) {
  r.x = one_of<F>();
  r.y = one_of<F>();
  r.z = zero_of<F>();
}
template <class F> GS_HD void jac_from_aff(Jac<F>& r, const Aff<F>& p) {
  if (aff_is_inf(p)) {
    jac_set_inf(r);
  } else {
    r.x = p.x;
    r.y = p.y;
    r.z = one_of<F>();
  }
}
template <class F> GS_HD void aff_neg(Aff<F>& r, const Aff<F>& p) {
  r.x = p.x;
  r.y = neg(p.y);
}

// All coordinates are stored normalised (N); the comments give the limb-growth A of
// the lazy intermediates (gs_tower.cuh header).

// dbl-2009-l: 2M + 5S
template <class F> GS_JAC void jac_dbl(Jac<F>& r, const Jac<F>& p) {
  F a = sqr(p.x), b = sqr(p.y), c = sqr(b);
  F d = norm(dbl(sub(sub(sqr_l2(add(p.x, b)), a), c)));  // 2 * 3 = 6 -> N
  F e = norm(add(dbl(a), a));                            // 3 -> N
  F f = sqr(e);
  F z3 = norm(dbl(mul(p.y, p.z)));
  F x3 = norm(sub(f, dbl(d)));                           // 3 -> N
  F c8 = dbl(norm(dbl(dbl(c))));                         // 8c with A = 2
  r.y = norm(sub(mul(e, norm(sub(d, x3))), c8));
  r.x = x3;
  r.z = z3;  // Z = 0 (exact) stays exactly 0
}

// madd-2007-bl with full edge-case handling: r = p + q (q affine)
template <class F> GS_JAC void jac_madd(Jac<F>& r, const Jac<F>& p, const Aff<F>& q) {
  if (aff_is_inf(q)) {
    r = p;
    return;
  }
  if (is_zero_limbs(p.z)) {
    r.x = q.x;
    r.y = q.y;
    r.z = one_of<F>();
    return;
  }
  F z1z1 = sqr(p.z);
  F u2 = mul(q.x, z1z1);
  F s2 = mul(mul(q.y, p.z), z1z1);
  F h = norm(sub(u2, p.x));
  F rr = sub(s2, p.y);
  if (is_zero(h)) {
    if (is_zero(rr)) {
      jac_dbl(r, p);
    } else {
      jac_set_inf(r);
    }
    return;
  }
  rr = norm(dbl(rr));  // 4 -> N
  F hh = sqr(h);
  F i = norm(dbl(dbl(hh)));
  F j = mul(h, i);
  F v = mul(p.x, i);
  F x3 = norm(sub(sub(sqr(rr), j), dbl(v)));  // 4 -> N
  F y3 = norm(sub(mul(rr, norm(sub(v, x3))), dbl(mul(p.y, j))));
  F z3 = norm(sub(sub(sqr_l2(add(p.z, h)), z1z1), hh));
  r.x = x3;
  r.y = y3;
  r.z = z3;
}

// add-2007-bl with full edge-case handling: r = p + q
template <class F> GS_JAC void jac_add(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
  if (is_zero_limbs(q.z)) {
    r = p;
    return;
  }
  if (is_zero_limbs(p.z)) {
    r = q;
    return;
  }
  F z1z1 = sqr(p.z), z2z2 = sqr(q.z);
  F u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
  F s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
  F h = norm(sub(u2, u1));
  F rr = sub(s2, s1);
  if (is_zero(h)) {
    if (is_zero(rr)) {
      jac_dbl(r, p);
    } else {
      jac_set_inf(r);
    }
    return;
  }
  rr = norm(dbl(rr));
  F i = sqr(norm(dbl(h)));
  F j = mul(h, i);
  F v = mul(u1, i);
  F x3 = norm(sub(sub(sqr(rr), j), dbl(v)));
  F y3 = norm(sub(mul(rr, norm(sub(v, x3))), dbl(mul(s1, j))));
  F z3 = mul(norm(sub(sub(sqr_l2(add(p.z, q.z)), z1z1), z2z2)), h);
  r.x = x3;
  r.y = y3;
  r.z = z3;
}

// ---- fast in-place forms for the scalar-multiplication loops ----------------------------------------------------------
// G1 / G2 on the device: the whole point operation is ONE generated subroutine on fixed registers (gs_pointops_asm.h:
// G1 straight-line, no operand moves, no spills, no memory instruction; G2 the compact form around the shared Fp2
// multiplier).  The subroutine is the generic branch of the formulas; the edge cases of the addition stay in C++: an
// identity operand is tested before the call, H = 0 mod p (P = +-Q) by the subroutine itself (56-bit filter, early
// return with the operands untouched) -- then the C++ jac_madd runs.  The CPU twin and builds without the asm calls
// take the C++ formulas.
template <class F> GS_HD_NOINLINE void jac_madd_edge(Jac<F>& r, const Jac<F>& p, const Aff<F>& q) { jac_madd(r, p, q); }
template <class F> GS_HD void jac_dbl_ip(Jac<F>& r) { jac_dbl(r, r); }
template <class F> GS_HD void jac_madd_ip(Jac<F>& r, const Aff<F>& q) { jac_madd(r, r, q); }
#if defined(GS_POINT_ASM)
template <class C> GS_HD void jac_dbl_ip(Jac<Fq<C>>& r) {
  // (Z = 0 stays exactly 0: products with exact zero limbs are exact zeros)
  if constexpr (C::L == 14)
    g1_dbl_call_14<C>(r.x.v, r.y.v, r.z.v);
  else
    g1_dbl_call_10<C>(r.x.v, r.y.v, r.z.v);
}
template <class C> GS_HD void jac_madd_ip(Jac<Fq<C>>& r, const Aff<Fq<C>>& q) {
  // The subroutine tests H = 0 ITSELF (56-bit filter, false alarms 2^-35 per lane) and returns at once, flag = 1, with
  // both operands untouched when any active lane may have P = +-Q or has an identity operand (`edge`).  So nothing but
  // the two operands is live across the call -- `r` never has its address taken (a running value that is
  // only reachable through a reference is memory at every step): the out-of-line routine works on copies made in the
  // rare branch.
  Aff<Fq<C>> qq = q;
  int32_t flag, edge = (aff_is_inf(q) || is_zero_limbs(r.z)) ? 1 : 0;
  // (the call is UNCONDITIONAL: the subroutine returns at once when any active lane has `edge` set)
  if constexpr (C::L == 14)
    g1_madd_call_14<C>(r.x.v, r.y.v, r.z.v, qq.x.v, qq.y.v, edge, flag);
  else
    g1_madd_call_10<C>(r.x.v, r.y.v, r.z.v, qq.x.v, qq.y.v, edge, flag);
  if (flag) {  // rare: the first addition of a lane, identity table entries, P = +-Q
    // (limb by limb: a whole-struct copy FROM `r` is a memcpy from its address, and hipcc then keeps the running point
    // in the private segment across the entire loop -- 11 loads before and 11 stores after every doubling call)
    Jac<Fq<C>> tp, tr;
    Aff<Fq<C>> tq;
#pragma unroll
    for (int i = 0; i < C::L; i++) {
      tp.x.v[i] = r.x.v[i], tp.y.v[i] = r.y.v[i], tp.z.v[i] = r.z.v[i];
      tq.x.v[i] = qq.x.v[i], tq.y.v[i] = qq.y.v[i];
    }
    jac_madd_edge(tr, tp, tq);
#pragma unroll
    for (int i = 0; i < C::L; i++) r.x.v[i] = tr.x.v[i], r.y.v[i] = tr.y.v[i], r.z.v[i] = tr.z.v[i];
  }
}
#if defined(GS_POINT_ASM_G2) && !defined(GS_POINT_ASM_G2_STRAIGHT)
// G2, compact form (gen_pointops_asm.py, Prog2c): the subroutine owns the data movement around the SHARED Fp2 multiplier
// subroutines, parks what does not fit in AGPRs, takes the addend in AGPRs and tests H = 0 itself -- it returns at once
// with flag = 1 and both operands untouched when any lane may have P = +-Q, so NOTHING but the two operands is live here
// across the call (copies of both kept for a test after the call cost hipcc the registers of the table entry it had just
// requested for the next step: k_fix.g2 12.3 -> 14.7 ms with the test outside).
template <class C> GS_HD void jac_dbl_ip(Jac<Fp2<C>>& r) {
  if constexpr (C::L == 14)
    g2_dbl_call_14<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v);
  else
    g2_dbl_call_10<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v);
}
template <class C> GS_HD void jac_madd_ip(Jac<Fp2<C>>& r, const Aff<Fp2<C>>& q) {
  Aff<Fp2<C>> qq = q;
  int32_t flag = 1;  // lanes that skip the call (either operand the identity) take the C++ addition as well
  if (!(aff_is_inf(q) || is_zero_limbs(r.z))) {
    if constexpr (C::L == 14)
      g2_madd_call_14<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v, qq.x.c0.v, qq.x.c1.v, qq.y.c0.v, qq.y.c1.v, flag);
    else
      g2_madd_call_10<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v, qq.x.c0.v, qq.x.c1.v, qq.y.c0.v, qq.y.c1.v, flag);
  }
  if (flag) {  // rare: first addition of a lane, identity table entries, P = +-Q (false alarms 2^-35 per lane)
    Jac<Fp2<C>> tp = r, tr;
    Aff<Fp2<C>> tq = qq;
    jac_madd_edge(tr, tp, tq);
    r = tr;
  }
}
#elif defined(GS_POINT_ASM_G2_STRAIGHT)
// G2: the same, the subroutines park what does not fit 256 VGPRs in AGPRs themselves (gen_pointops_asm.py, Prog2).
// MEASURED AND NOT SHIPPED (round 4, gpurun_out r4e / r4f on 2^16 PPE): bit-exact on every forced shape, but
// k_var_multi8w5x2.g2 40.0 -> 41.5 ms and k_fix.g2 12.2 -> 17.8 ms.  The straight-line G2 addition is 114 KB of code
// and the doubling 68 KB -- against a 64 KB instruction cache shared by two CUs -- while hipcc's version keeps
// calling the SAME 10.6 KB Fp2 multiplier: fewer instructions (14.2 k against ~17 k per addition) lose to instruction
// fetch.  The G1 pair (42 + 26 KB) stays on the winning side (-11..13 %).  -DGS_POINT_ASM_G2_STRAIGHT brings this back.
template <class C> GS_HD void jac_dbl_ip(Jac<Fp2<C>>& r) {
  if constexpr (C::L == 14)
    g2_dbl_call_14<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v);
  else
    g2_dbl_call_10<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v);
}
template <class C> GS_HD void jac_madd_ip(Jac<Fp2<C>>& r, const Aff<Fp2<C>>& q) {
  const Jac<Fp2<C>> p0 = r;
  Aff<Fp2<C>> qq = q;
  const bool edge = aff_is_inf(q) || is_zero_limbs(r.z);
  int32_t h[4];
  if constexpr (C::L == 14)
    g2_madd_call_14<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v, qq.x.c0.v, qq.x.c1.v, qq.y.c0.v, qq.y.c1.v, h);
  else
    g2_madd_call_10<C>(r.x.c0.v, r.x.c1.v, r.y.c0.v, r.y.c1.v, r.z.c0.v, r.z.c1.v, qq.x.c0.v, qq.x.c1.v, qq.y.c0.v, qq.y.c1.v, h);
  if (edge || (maybe_zero_limbs01<C>(h[0], h[1]) && maybe_zero_limbs01<C>(h[2], h[3]))) {
    Jac<Fp2<C>> tp = p0, tr;
    Aff<Fp2<C>> tq = q;
    jac_madd_edge(tr, tp, tq);
    r = tr;
  }
}
#endif
#endif

template <class F> GS_HD void jac_neg(Jac<F>& r, const Jac<F>& p) {
  r.x = p.x;
  r.y = neg(p.y);
  r.z = p.z;
}

// normalise with a known z^-1
template <class F> GS_HD void jac_to_aff_zinv(Aff<F>& r, const Jac<F>& p, const F& zinv) {
  if (is_zero_limbs(p.z)) {
    r.x = zero_of<F>();
    r.y = zero_of<F>();
    return;
  }
  F zi2 = sqr(zinv);
  r.x = mul(p.x, zi2);
  r.y = mul(mul(p.y, zi2), zinv);
}
template <class F> GS_HD void jac_to_aff(Aff<F>& r, const Jac<F>& p) {
  F zi = inv(p.z);
  jac_to_aff_zinv(r, p, zi);
}

// Signed fixed-window (w = 4) scalar multiplication: digits in [-8, 8).
// `k` is a canonical (non-Montgomery) scalar.  tab[i] = (i+1) * P, i = 0..7.
template <class F> GS_HD_NOINLINE void smul_build_table(Jac<F>* tab, const Aff<F>& p) {
  jac_from_aff(tab[0], p);
  // k even: kP = 2 (k/2)P ; k odd: kP = (k-1)P + P   (one doubling and one mixed-add call site)
#pragma unroll 1
  for (int k = 2; k <= 8; k++) {
    if (k & 1)
      jac_madd(tab[k - 1], tab[k - 2], p);
    else
      jac_dbl(tab[k - 1], tab[k / 2 - 1]);
  }
}

// Effective-affine tables ("global Z", the trick of libsecp256k1's ecmult).  On an a = 0 curve neither the doubling
// nor the addition formulas involve b, so M Jacobian entries (X_i, Y_i, Z_i) can be rescaled to ONE common
// denominator Zc = prod Z_i and read as AFFINE points (X_i s_i^2, Y_i s_i^3), s_i = Zc / Z_i, of the isomorphic curve
// y^2 = x^3 + b Zc^6.  The whole scalar multiplication then runs there with MIXED additions (11 instead of 16 Fq
// multiplications in G1, 29 instead of 43 in G2) for 7 multiplications per entry and no inversion; the result maps
// back with Z <- Z * zback.  In G2 the common denominator is made an element of Fq (Zc times its conjugate) so that
// psi, which conjugates Z, maps the isomorphic curve to itself.  Entries at infinity become the affine identity (0, 0).
template <class C> GS_HD Fq<C> gz_adjust(const Fq<C>&) { return fq_one<C>(); }
template <class C> GS_HD Fp2<C> gz_adjust(const Fp2<C>& zc) { return conj(zc); }
// (helpers of the loop below: limb-wise loads into locals that never have their address taken by an out-of-line call --
// a whole-struct copy from memory, or a value handed to is_zero()'s exact test by reference, makes hipcc keep the value in
// the private segment and move it 16 bytes at a time with a full wait in between)
template <class F> GS_HD void ld_limbs(F& d, const F* s) {
  constexpr int NW = (int)(sizeof(F) / sizeof(limb_t));
  const limb_t* w = reinterpret_cast<const limb_t*>(s);
#pragma unroll
  for (int q = 0; q < NW; q++) reinterpret_cast<limb_t*>(&d)[q] = w[q];
}
template <class C> GS_HD bool gz_maybe_zero(const Fq<C>& a) { return maybe_zero_limbs01<C>(a.v[0], a.v[1]); }
template <class C> GS_HD bool gz_maybe_zero(const Fp2<C>& a) { return gz_maybe_zero<C>(a.c0) && gz_maybe_zero<C>(a.c1); }
template <class C, class F> GS_HD bool gz_is_zero(const F& z) {
  if (!gz_maybe_zero<C>(z)) return false;  // 56-bit filter, inline
  F t = z;                                 // the exact test (out of line, by reference) works on a copy made HERE
  return is_zero(t);
}
template <class C, class F> GS_HD_NOINLINE void table_global_z(Aff<F>* aff, const Jac<F>* tab, int M, F& zback) {
  // Both passes read every entry ONCE from the lane's workspace (HBM): each entry is requested one iteration ahead, so
  // that its latency hides behind the multiplications of the current one -- at one wave per SIMD a load issued where it
  // is needed is 1-2 us of nothing, 2 x 128 times per table build (round 4: 13 % of the wave cycles of the 8-term G1
  // lanes were s_waitcnt, most of them here).
  F acc = one_of<F>();
  F zn;
  ld_limbs(zn, &tab[0].z);
  for (int i = 0; i < M; i++) {
    const F z = zn;
    if (i + 1 < M) ld_limbs(zn, &tab[i + 1].z);
    aff[i].x = acc;  // prefix product, replaced below
    if (!gz_is_zero<C>(z)) acc = mul(acc, z);
  }
  F adj = gz_adjust<C>(acc);
  zback = mul(acc, adj);
  F suf = adj;
  F ex, ey, ez, pre;
  ld_limbs(ex, &tab[M - 1].x), ld_limbs(ey, &tab[M - 1].y), ld_limbs(ez, &tab[M - 1].z), ld_limbs(pre, &aff[M - 1].x);
  for (int i = M - 1; i >= 0; i--) {
    const F cx = ex, cy = ey, cz = ez, cp = pre;
    if (i > 0) ld_limbs(ex, &tab[i - 1].x), ld_limbs(ey, &tab[i - 1].y), ld_limbs(ez, &tab[i - 1].z), ld_limbs(pre, &aff[i - 1].x);
    if (gz_is_zero<C>(cz)) {
      aff[i].x = zero_of<F>();
      aff[i].y = zero_of<F>();
      continue;
    }
    F sc = mul(cp, suf);
    suf = mul(suf, cz);
    F s2 = sqr(sc);
    aff[i].x = mul(cx, s2);
    aff[i].y = mul(cy, mul(s2, sc));
  }
}

// Recode a canonical scalar of NB bits into ceil((NB+1)/4) signed digits
// d_i in [-8, 8) with sum d_i 16^i = k.  Digits are produced on the fly from
// the top: returns digit i given the running scheme  d_i = ((k >> 4i) & 15) + carry_i.
// To stay branch-light we precompute them into a small byte array.
template <class M> GS_HD void recode_w4(int8_t* dg, int nd, const Fe<M>& k) {
  uint32_t carry = 0;
  for (int i = 0; i < nd; i++) {
    uint32_t v = get_bits(k, 4 * i, 4) + carry;  // 0..16
    if (4 * i >= M::N * 32) v = carry;
    if (v >= 8) {
      dg[i] = (int8_t)((int)v - 16);
      carry = 1;
    } else {
      dg[i] = (int8_t)v;
      carry = 0;
    }
  }
}

// r = k * P  (k canonical scalar in Fr's limb type M)
template <class F, class M> GS_HD_NOINLINE void jac_smul(Jac<F>& rout, const Aff<F>& p, const Fe<M>& k) {
  Jac<F> r;  // local running sum (see jac_msm_straus_at)
  constexpr int ND = (M::BITS + 3) / 4 + 1;  // one spare digit for the signed carry
  Jac<F> tab[8];
  Aff<F> at[8];
  F zback;
  int8_t dg[ND];
  smul_build_table(tab, p);
  table_global_z<typename M::Curve>(at, tab, 8, zback);
  recode_w4<M>(dg, ND, k);
  jac_set_inf(r);
  for (int i = ND - 1; i >= 0; i--) {
    if (i != ND - 1) {
#pragma unroll 1
      for (int d4 = 0; d4 < 4; d4++) jac_dbl(r, r);
    }
    int d = dg[i];
    if (d != 0) {
      int a = d < 0 ? -d : d;
      Aff<F> t = at[a - 1];
      if (d < 0) t.y = neg(t.y);
      jac_madd(r, r, t);
    }
  }
  r.z = mul(r.z, zback);
  rout = r;
}

// ---------------------------------------------------------------------------
// Endomorphism-accelerated scalar multiplication (BLS12-381; C::HAS_ENDO).
// Valid for points of the prime-order subgroups (as every arkworks-deserialised
// point is).  G1: GLV, k = k1 + k2*lambda with lambda = x^2 - 1 a cube root of unity
// mod r and phi(X,Y,Z) = (beta X, Y, Z); because lambda ~ 2^128 ~ sqrt(r) the
// decomposition is plain division with remainder.  G2: psi = twist o Frobenius o
// untwist acts as [x] (x < 0), so the base-|x| digits d0..d3 of k give
//   k Q = d0 Q - d1 psi(Q) + d2 psi^2(Q) - d3 psi^3(Q)     (64-bit sub-scalars).
// Both share ONE 8-entry table; the endomorphism is applied to the looked-up
// Jacobian entry.  Doublings drop from 256 to 132 (G1) / 68 (G2).
// ---------------------------------------------------------------------------
// q = n / d, rem = n % d for little-endian u32 limb arrays (restoring division, NN <= 8, ND <= 4)
template <int NN, int ND> GS_HD void limb_divmod(uint32_t* q, uint32_t* rem, const uint32_t* n, const uint32_t* d) {
  uint32_t r[ND + 1];
  for (int i = 0; i <= ND; i++) r[i] = 0;
  for (int i = 0; i < NN; i++) q[i] = 0;
  for (int b = NN * 32 - 1; b >= 0; b--) {
    // r = (r << 1) | bit
    for (int i = ND; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
    r[0] = (r[0] << 1) | ((n[b >> 5] >> (b & 31)) & 1u);
    // t = r - d
    uint32_t t[ND + 1];
    uint32_t br = 0;
    for (int i = 0; i <= ND; i++) {
      uint64_t x = (uint64_t)r[i] - (i < ND ? d[i] : 0u) - br;
      t[i] = (uint32_t)x;
      br = (uint32_t)(x >> 63);
    }
    if (!br) {
      for (int i = 0; i <= ND; i++) r[i] = t[i];
      q[b >> 5] |= 1u << (b & 31);
    }
  }
  for (int i = 0; i < ND; i++) rem[i] = r[i];
}
// The same division by a COMPILE-TIME divisor D (the GLV / GLS decompositions divide by lambda = x^2 - 1 and by |x|): Barrett,
// q^ = floor(n mu / 2^(32 NN)) with mu = floor(2^(32 NN) / D) evaluated by the compiler, then at most two corrections
// (q - 2 <= q^ <= q).  ~150 instructions where the bit-by-bit loop above takes ~25 per bit x 256 bits -- round 4's phase
// stamps put the digit preparation at 5 % of a G1 Straus lane (367 k of 7.3 M cycles per output, tools/straus_phases.py).
template <int NN, int ND> struct BarrettMu {
  uint32_t v[NN + 1];
};
template <int NN, int ND, class DT> constexpr BarrettMu<NN, ND> barrett_mu(const DT& d) {
  BarrettMu<NN, ND> m{};
  uint32_t r[ND + 1] = {};
  for (int b = 32 * NN; b >= 0; b--) {
    for (int i = ND; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
    r[0] = (r[0] << 1) | (b == 32 * NN ? 1u : 0u);
    uint32_t t[ND + 1] = {};
    uint32_t br = 0;
    for (int i = 0; i <= ND; i++) {
      uint64_t x = (uint64_t)r[i] - (i < ND ? (uint64_t)d[i] : 0u) - br;
      t[i] = (uint32_t)x;
      br = (uint32_t)(x >> 63);
    }
    if (!br) {
      for (int i = 0; i <= ND; i++) r[i] = t[i];
      m.v[b >> 5] |= 1u << (b & 31);
    }
  }
  return m;
}
// D: a type with a static constexpr uint32_t array `d[ND]` (little-endian limbs, top limb non-zero)
template <int NN, int ND, class D> GS_HD void limb_divmod_const(uint32_t* q, uint32_t* rem, const uint32_t* n) {
  constexpr BarrettMu<NN, ND> MU = barrett_mu<NN, ND>(D::d);
  constexpr int NM = NN - ND + 2;  // limbs of mu that can be non-zero (mu < 2^(32 (NN - ND) + 33))
  static_assert(NM <= NN + 1, "divisor too short");
  // t = n * mu, limbs NN .. NN + NM - 1 are q^ (the low limbs only matter through their carries)
  uint32_t prod[NN + NM];
  for (int i = 0; i < NN + NM; i++) prod[i] = 0;
  for (int i = 0; i < NN; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < NM; j++) {
      uint64_t t = (uint64_t)n[i] * MU.v[j] + prod[i + j] + carry;
      prod[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    prod[i + NM] = (uint32_t)carry;
  }
  uint32_t qh[NM];
  for (int j = 0; j < NM; j++) qh[j] = prod[NN + j];
  // r = n - q^ D on ND + 1 limbs (0 <= r < 3 D)
  uint32_t qd[ND + 1];
  for (int i = 0; i <= ND; i++) qd[i] = 0;
  for (int i = 0; i < NM && i <= ND; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < ND && i + j <= ND; j++) {
      uint64_t t = (uint64_t)qh[i] * D::d[j] + qd[i + j] + carry;
      qd[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    if (i + ND <= ND) qd[i + ND] += (uint32_t)carry;
  }
  uint32_t r[ND + 1];
  {
    uint32_t br = 0;
    for (int i = 0; i <= ND; i++) {
      uint64_t x = (uint64_t)(i < NN ? n[i] : 0u) - qd[i] - br;
      r[i] = (uint32_t)x;
      br = (uint32_t)(x >> 63);
    }
  }
  for (int it = 0; it < 2; it++) {  // while (r >= D) { r -= D; q^++; }
    uint32_t t[ND + 1];
    uint32_t br = 0;
    for (int i = 0; i <= ND; i++) {
      uint64_t x = (uint64_t)r[i] - (i < ND ? D::d[i] : 0u) - br;
      t[i] = (uint32_t)x;
      br = (uint32_t)(x >> 63);
    }
    if (!br) {
      for (int i = 0; i <= ND; i++) r[i] = t[i];
      uint32_t c = 1;
      for (int j = 0; j < NM; j++) {
        uint64_t x = (uint64_t)qh[j] + c;
        qh[j] = (uint32_t)x;
        c = (uint32_t)(x >> 32);
      }
    }
  }
  for (int i = 0; i < NN; i++) q[i] = i < NM ? qh[i] : 0u;
  for (int i = 0; i < ND; i++) rem[i] = r[i];
}
template <class C> struct DivLambda {
  static constexpr const uint32_t (&d)[4] = C::LAMBDA;
};
template <class C> struct DivXabs {
  static constexpr const uint32_t (&d)[2] = C::XABS_LIMBS;
};
// signed w=4 digits of an NL-limb value; nd = 8*NL + 1 digits
template <int NL> GS_HD void recode_w4_limbs(int8_t* dg, const uint32_t* k) {
  uint32_t carry = 0;
  for (int i = 0; i < 8 * NL + 1; i++) {
    uint32_t nib = i < 8 * NL ? ((k[i >> 3] >> ((i & 7) * 4)) & 15u) : 0u;
    uint32_t v = nib + carry;
    if (v >= 8) {
      dg[i] = (int8_t)((int)v - 16);
      carry = 1;
    } else {
      dg[i] = (int8_t)v;
      carry = 0;
    }
  }
}

// signed width-W digits (in [-2^(W-1), 2^(W-1))) of an NL-limb value; (32 NL + W - 1) / W + 1 of them
template <int NL, int W> GS_HD void recode_w_limbs(int8_t* dg, const uint32_t* k) {
  constexpr int ND = (32 * NL + W - 1) / W + 1;
  uint32_t carry = 0;
  for (int i = 0; i < ND; i++) {
    int bit = i * W;
    uint32_t v = 0;
    if (bit < 32 * NL) {
      v = k[bit >> 5] >> (bit & 31);
      if ((bit & 31) + W > 32 && (bit >> 5) + 1 < NL) v |= k[(bit >> 5) + 1] << (32 - (bit & 31));
      v &= (1u << W) - 1u;
    }
    v += carry;
    if (v >= (1u << (W - 1))) {
      dg[i] = (int8_t)((int)v - (1 << W));
      carry = 1;
    } else {
      dg[i] = (int8_t)v;
      carry = 0;
    }
  }
}
// tab[i] = (i + 1) P, i = 0 .. NE-1 (k even: 2 (k/2)P; k odd: (k-1)P + P)
// The chain runs on a value held in registers and updated in place (the G1 / G2 register-only subroutines on the
// device): an odd multiple continues from the entry just produced, an even one reloads (k/2) P.
template <class F> GS_HD_NOINLINE void smul_build_table_n(Jac<F>* tab, const Aff<F>& p, int ne) {
  Jac<F> cur;
  jac_from_aff(cur, p);
  tab[0] = cur;
#pragma unroll 1
  for (int k = 2; k <= ne; k++) {
    if (k & 1) {
      jac_madd_ip(cur, p);  // (k - 1) P is what the previous step left in `cur`
    } else {
      Jac<F> t = tab[k / 2 - 1];
      cur = t;
      jac_dbl_ip(cur);
    }
    tab[k - 1] = cur;
  }
}

// out[i] = (i + 1) P in AFFINE form, i < ne: Jacobian chain into `stage`, one inversion for all entries (Montgomery's
// trick).  For P of prime order and ne < r no multiple is the identity unless P is -- but this table is built from the
// VERIFIER's commitment inputs (k_tab_build), i.e. from untrusted boundary data: a low-order or non-subgroup point can
// make some multiple the identity (Z = 0).  Such entries are skipped in the prefix product (as k_red does) and come out
// as the affine identity (0, 0), so one bad base cannot poison the shared inversion of its table.
template <class F> GS_HD_NOINLINE void smul_affine_table(Aff<F>* out, Jac<F>* stage, const Aff<F>& p, int ne) {
  if (aff_is_inf(p)) {
    for (int i = 0; i < ne; i++) out[i].x = zero_of<F>(), out[i].y = zero_of<F>();
    return;
  }
  smul_build_table_n(stage, p, ne);
  F acc = one_of<F>();
  for (int i = 0; i < ne; i++) {
    out[i].x = acc;  // prefix product of the non-zero Z before entry i
    if (!is_zero_limbs(stage[i].z)) acc = mul(acc, stage[i].z);
  }
  F iv = inv(acc);
  for (int i = ne - 1; i >= 0; i--) {
    if (is_zero_limbs(stage[i].z)) {
      out[i].x = zero_of<F>();
      out[i].y = zero_of<F>();
      continue;
    }
    F zi = mul(out[i].x, iv);  // 1 / Z_i
    iv = mul(iv, stage[i].z);
    F z2 = sqr(zi);
    out[i].x = mul(stage[i].x, z2);
    out[i].y = mul(stage[i].y, mul(z2, zi));
  }
}

// ---- lattice decomposition (BN curves: no eigenvalue of size sqrt(r) / r^(1/4) exists, so the sub-scalars come from
// Babai rounding against a reduced basis B of {v : sum v_j eig^j = 0 mod r}; gen_params.py derives and checks B and
// G_i = floor(2^256 |(B^-1)_0i|)).  c_i = floor(k G_i / 2^256) (sign GS_i), k_j = [j = 0] k - sum_i c_i B_ij.
template <int NA, int NB> GS_HD void mp_mul(uint32_t* out, const uint32_t* a, const uint32_t* b) {
  for (int i = 0; i < NA + NB; i++) out[i] = 0;
  for (int i = 0; i < NA; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < NB; j++) {
      uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    out[i + NB] = (uint32_t)carry;
  }
}
// acc (W words, two's complement) -= or += a (NA <= W words, non-negative)
template <int W, int NA> GS_HD void mp_acc(uint32_t* acc, const uint32_t* a, bool add) {
  uint64_t c = add ? 0 : 1;  // a + (~b + 1) = a - b
  for (int i = 0; i < W; i++) {
    uint32_t w = i < NA ? a[i] : 0u;
    if (!add) w = ~w;
    uint64_t t = (uint64_t)acc[i] + w + c;
    acc[i] = (uint32_t)t;
    c = t >> 32;
  }
}
template <int D, int GW, int BW, int NL>
GS_HD void endo_lattice(uint32_t (*mag)[NL], uint8_t* sgn, const uint32_t* k, const uint32_t (*G)[GW], const uint8_t* GS,
                        const uint32_t (*B)[D][BW], const uint8_t (*BS)[D]) {
  constexpr int W = 10;
  static_assert(GW + BW < W && NL <= W, "decomposition width");
  uint32_t c[D][GW];
  for (int i = 0; i < D; i++) {
    uint32_t t[8 + GW], g[GW];
    for (int w = 0; w < GW; w++) g[w] = G[i][w];
    mp_mul<8, GW>(t, k, g);
    for (int w = 0; w < GW; w++) c[i][w] = t[8 + w];
  }
  for (int j = 0; j < D; j++) {
    uint32_t acc[W];
    for (int w = 0; w < W; w++) acc[w] = (j == 0 && w < 8) ? k[w] : 0u;
    for (int i = 0; i < D; i++) {
      uint32_t t[GW + BW], b[BW];
      for (int w = 0; w < BW; w++) b[w] = B[i][j][w];
      mp_mul<GW, BW>(t, c[i], b);
      bool neg_prod = (GS[i] != 0) != (BS[i][j] != 0);
      mp_acc<W, GW + BW>(acc, t, neg_prod);  // acc -= (+-) c_i B_ij
    }
    bool negative = (acc[W - 1] >> 31) != 0;
    if (negative) {
      uint64_t cy = 1;
      for (int w = 0; w < W; w++) {
        uint64_t t = (uint64_t)(~acc[w]) + cy;
        acc[w] = (uint32_t)t;
        cy = t >> 32;
      }
    }
    sgn[j] = negative ? 1 : 0;
    for (int w = 0; w < NL; w++) mag[j][w] = acc[w];
  }
}

template <class C> GS_HD_NOINLINE void jac_smul_endo(Jac<Fq<C>>& rout, const Aff<Fq<C>>& p, const Fr<C>& k) {
  Jac<Fq<C>> r;
  uint32_t kk[8], q[8], k1[4], k2[4], lam[4];
  for (int i = 0; i < 8; i++) kk[i] = k.v[i];
  for (int i = 0; i < 4; i++) lam[i] = C::LAMBDA[i];
  limb_divmod_const<8, 4, DivLambda<C>>(q, k1, kk);
  (void)lam;
  for (int i = 0; i < 4; i++) k2[i] = q[i];
  Jac<Fq<C>> tab[8];
  Aff<Fq<C>> at[8];
  Fq<C> zback;
  int8_t d1[33], d2[33];
  smul_build_table(tab, p);
  table_global_z<C>(at, tab, 8, zback);
  recode_w4_limbs<4>(d1, k1);
  recode_w4_limbs<4>(d2, k2);
  Fq<C> beta;
  for (int i = 0; i < C::L; i++) beta.v[i] = C::BETA_28[i];
  jac_set_inf(r);
  for (int i = 32; i >= 0; i--) {
    if (i != 32) {
#pragma unroll 1
      // (the C++ formulas, not the register-only subroutines of the Straus lanes: with the table a dynamically indexed
      // private array the subroutine's fixed operand registers cost more than they save here -- measured at 2^12,
      // k_var.g1 1.88 ms against 2.13 ms)
      for (int d4 = 0; d4 < 4; d4++) jac_dbl(r, r);
    }
    int a = d1[i];
    if (a != 0) {
      Aff<Fq<C>> t = at[(a < 0 ? -a : a) - 1];
      if (a < 0) t.y = neg(t.y);
      jac_madd(r, r, t);
    }
    int b = d2[i];
    if (b != 0) {
      Aff<Fq<C>> t = at[(b < 0 ? -b : b) - 1];
      t.x = mul(t.x, beta);
      if (b < 0) t.y = neg(t.y);
      jac_madd(r, r, t);
    }
  }
  r.z = mul(r.z, zback);
  rout = r;
}

template <class C> GS_HD Fp2<C> fp2_const28(const int32_t (*c)[C::L]) {
  Fp2<C> r;
  for (int i = 0; i < C::L; i++) {
    r.c0.v[i] = c[0][i];
    r.c1.v[i] = c[1][i];
  }
  return r;
}
// the endomorphisms on affine entries of an effective-affine table (same constants as on Jacobian X, Y)
template <class C> GS_HD void endo_apply(Aff<Fq<C>>& t, int s) {
  if (s == 1) {
    Fq<C> beta;
    for (int i = 0; i < C::L; i++) beta.v[i] = C::BETA_28[i];
    t.x = mul(t.x, beta);
  }
}
template <class C> GS_HD void endo_apply(Aff<Fp2<C>>& t, int s) {
  if (aff_is_inf(t)) return;  // (0, 0) stays the identity flag
  if constexpr (C::IS_BN) {
    // psi^s on the D-type twist = the twist Frobenius of the Miller loop: conjugate s times, scale by xi^(k (p^s-1)/6)
    if (s == 0) return;
    Fp2<C> x = (s & 1) ? conj(t.x) : t.x, y = (s & 1) ? conj(t.y) : t.y;
    t.x = mul(x, frob_coeff<C>(s, 2));
    t.y = mul(y, frob_coeff<C>(s, 3));
    return;
  } else if (s == 1) {
    t.x = mul(conj(t.x), fp2_const28<C>(C::PSI_X_28));
    t.y = mul(conj(t.y), fp2_const28<C>(C::PSI_Y_28));
  } else if (s == 2) {
    Fq<C> nx, ny;
    for (int l = 0; l < C::L; l++) {
      nx.v[l] = C::PSI2_X_28[l];
      ny.v[l] = C::PSI2_Y_28[l];
    }
    t.x = mul_fp(t.x, nx);
    t.y = mul_fp(t.y, ny);
  } else if (s == 3) {
    t.x = mul(conj(t.x), fp2_const28<C>(C::PSI3_X_28));
    t.y = mul(conj(t.y), fp2_const28<C>(C::PSI3_Y_28));
  }
}
template <class C> GS_HD_NOINLINE void jac_smul_endo(Jac<Fp2<C>>& rout, const Aff<Fp2<C>>& p, const Fr<C>& k) {
  Jac<Fp2<C>> r;
  // base-|x| digits of k
  uint32_t n[8], q[8], xa[2], d[4][2];
  for (int i = 0; i < 8; i++) n[i] = k.v[i];
  xa[0] = C::XABS_LIMBS[0];
  xa[1] = C::XABS_LIMBS[1];
  for (int j = 0; j < 3; j++) {
    limb_divmod_const<8, 2, DivXabs<C>>(q, d[j], n);
    for (int i = 0; i < 8; i++) n[i] = q[i];
  }
  (void)xa;
  d[3][0] = n[0];
  d[3][1] = n[1];
  Jac<Fp2<C>> tab[8];
  Aff<Fp2<C>> at[8];
  Fp2<C> zback;
  int8_t dg[4][17];
  smul_build_table(tab, p);
  table_global_z<C>(at, tab, 8, zback);
  for (int j = 0; j < 4; j++) recode_w4_limbs<2>(dg[j], d[j]);
  jac_set_inf(r);
  for (int i = 16; i >= 0; i--) {
    if (i != 16) {
#pragma unroll 1
      for (int d4 = 0; d4 < 4; d4++) jac_dbl_ip(r);
    }
    for (int j = 0; j < 4; j++) {
      int a = dg[j][i];
      if (a == 0) continue;
      Aff<Fp2<C>> t = at[(a < 0 ? -a : a) - 1];
      bool negate = (a < 0) != ((j & 1) != 0);  // bases: +Q, -psi Q, +psi^2 Q, -psi^3 Q
      endo_apply<C>(t, j);
      if (negate) t.y = neg(t.y);
      jac_madd_ip(r, t);
    }
  }
  r.z = mul(r.z, zback);
  rout = r;
}

template <class C, class F, int TMAX>
GS_HD void jac_msm_straus_at(Jac<F>& r, const Aff<F>* ps, const Fr<C>* ks, int nt, Aff<F>* at);
template <class C, class F, int TMAX> GS_HD void jac_msm_straus(Jac<F>& r, const Aff<F>* ps, const Fr<C>* ks, int nt) {
  Aff<F> at[TMAX * 8];
  jac_msm_straus_at<C, F, TMAX>(r, ps, ks, nt, at);
}
// dispatch: endomorphism path where the curve has one (BN curves: the one-term case of the joint routine, whose
// digit streams come from the lattice decomposition)
// ENDO = false: plain signed-window double-and-add (jac_smul), defined on ANY curve point like the reference's
// Com::scalar_mul (data_structures.rs:336-342); the endomorphism paths are only valid on the r-torsion.
template <class C, class F, bool ENDO = true> GS_HD void jac_smul_any(Jac<F>& r, const Aff<F>& p, const Fr<C>& k) {
  if constexpr (!ENDO)
    jac_smul(r, p, k);
  else if constexpr (C::HAS_ENDO && C::IS_BN)
    jac_msm_straus<C, F, 1>(r, &p, &k, 1);
  else if constexpr (C::HAS_ENDO)
    jac_smul_endo<C>(r, p, k);
  else
    jac_smul(r, p, k);
}

// ---------------------------------------------------------------------------
// Joint (Straus) multi-scalar multiplication  r = sum_t k_t P_t,  nt <= TMAX terms
// per lane: ONE doubling chain shared by all terms (and by their endomorphism
// sub-scalars), one 8-entry table per base.  Used for the Gamma-weighted inner
// products once the batch is large enough that fewer, longer lanes still fill the chip.
// ---------------------------------------------------------------------------
// Sub-scalar digit streams of one term (signed w = 4 digits, NS streams of ND digits) and the sign of each stream.
//   BLS12: G1 k = k1 + k2 lambda by division (2 x 33 digits); G2 base-|x| digits with the fixed sign pattern
//          +Q, -psi Q, +psi^2 Q, -psi^3 Q (4 x 17 digits)
//   BN   : lattice decomposition, signs per scalar: G1 2 x 41 digits (<= 160-bit slots), G2 4 x 25 (<= 96-bit)
template <class C, class F> struct EndoShape;
template <class C> struct EndoShape<C, Fq<C>> {
  static constexpr int NS = 2, NL = C::IS_BN ? 5 : 4, ND = 8 * NL + 1;
  static constexpr int nd(int w) { return (32 * NL + w - 1) / w + 1; }
};
template <class C> struct EndoShape<C, Fp2<C>> {
  static constexpr int NS = 4, NL = C::IS_BN ? 3 : 2, ND = 8 * NL + 1;
  static constexpr int nd(int w) { return (32 * NL + w - 1) / w + 1; }
};
template <class C, int W = 4> GS_HD void endo_digits(int8_t* dg, uint8_t* sgn, const Fr<C>& k, const Jac<Fq<C>>*) {
  typedef EndoShape<C, Fq<C>> E;
  constexpr int ND = E::nd(W);
  uint32_t kk[8];
  for (int i = 0; i < 8; i++) kk[i] = k.v[i];
  if constexpr (C::IS_BN) {
    uint32_t mag[2][E::NL];
    endo_lattice<2, 5, 4, E::NL>(mag, sgn, kk, C::GLV1_G, C::GLV1_GS, C::GLV1_B, C::GLV1_BS);
    for (int s = 0; s < 2; s++) recode_w_limbs<E::NL, W>(dg + ND * s, mag[s]);
  } else {
    uint32_t q[8], k1[4], k2[4], lam[4];
    for (int i = 0; i < 4; i++) lam[i] = C::LAMBDA[i];
    limb_divmod_const<8, 4, DivLambda<C>>(q, k1, kk);
    (void)lam;
    for (int i = 0; i < 4; i++) k2[i] = q[i];
    recode_w_limbs<4, W>(dg, k1);
    recode_w_limbs<4, W>(dg + ND, k2);
    sgn[0] = sgn[1] = 0;
  }
}
template <class C, int W = 4> GS_HD void endo_digits(int8_t* dg, uint8_t* sgn, const Fr<C>& k, const Jac<Fp2<C>>*) {
  typedef EndoShape<C, Fp2<C>> E;
  constexpr int ND = E::nd(W);
  uint32_t n[8];
  for (int i = 0; i < 8; i++) n[i] = k.v[i];
  if constexpr (C::IS_BN) {
    uint32_t mag[4][E::NL];
    endo_lattice<4, 7, 2, E::NL>(mag, sgn, n, C::GLS2_G, C::GLS2_GS, C::GLS2_B, C::GLS2_BS);
    for (int s = 0; s < 4; s++) recode_w_limbs<E::NL, W>(dg + ND * s, mag[s]);
  } else {
    uint32_t q[8], xa[2], d[4][2];
    xa[0] = C::XABS_LIMBS[0];
    xa[1] = C::XABS_LIMBS[1];
    for (int j = 0; j < 3; j++) {
      limb_divmod_const<8, 2, DivXabs<C>>(q, d[j], n);
      for (int i = 0; i < 8; i++) n[i] = q[i];
    }
    (void)xa;
    d[3][0] = n[0];
    d[3][1] = n[1];
    for (int j = 0; j < 4; j++) {
      recode_w_limbs<2, W>(dg + ND * j, d[j]);
      sgn[j] = (uint8_t)(j & 1);  // x < 0: psi acts as -|x|
    }
  }
}

// Tables and main loop are separate routines so that SEVERAL outputs over the same bases (the two proof elements of a
// side, the n columns of Gamma^T c in the verifier) share one table build: jac_straus_build once, jac_straus_run per
// output.  W = window width (signed digits, NE = 2^(W-1) entries per base): 4, or 5 when the build is shared.
// `at` = room for NE * TMAX table entries.  The main loop indexes it with the lane's own digit, so it must NOT live in
// the kernel's private frame: scratch is interleaved dword by dword across the 64 lanes of a wave, and 64 lanes reading
// 8 different entries pull 8 rows of 256 bytes for every 256 bytes they use.  The kernels pass a lane-contiguous
// global workspace instead (one entry = consecutive bytes of one lane).
// (Deliberately a function of its own.  Inlined into the kernel, the BLS12-381 G1 lanes drop from 257 to 253
// registers and become eligible for two waves per SIMD -- and the SAME lanes then take 27 ms instead of 17.7 ms at
// 2^16 when they are launched as 1024 waves, with k_fix / k_red of the same step slower too; measured on one box,
// profiles/r2/straus_occupancy.txt.)
// `tab` = room for the same number of JACOBIAN entries, the staging of the build.  It used to be a local array: 43 KB of
// private segment per lane for 8 G2 bases with 5-bit windows -- 3 GB of queue scratch for a full chip, which the
// runtime's scratch pool only grants while no other queue holds any (it then limits the kernel's resident waves: the
// same step took 1100-1700 ms instead of 320 ms after a small batch had run reductions on a side stream,
// profiles/r3/scratch_pool.txt).  The kernels pass a second region of the lane's global workspace.
#if defined(GS_DEBUG_STAMPS) && defined(__HIPCC__)
// diagnosis build (tools/build_variant.sh stamps -DGS_DEBUG_STAMPS): lane 0 of every wave adds shader-cycle deltas of
// the phases of the G1 Straus lanes to gs_dbg[]: 0 digits + top, 1 main loop, 2 inside the doubling subroutine calls,
// 3 inside the addition steps (table wait + endomorphism + call), 4 waves, 5 the table build (k_var_multi)
__device__ unsigned long long gs_dbg[16];
#endif
#if defined(GS_DEBUG_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define GS_STAMP_T() __builtin_amdgcn_s_memtime()
#define GS_STAMP_ADD(i, t) do { if ((threadIdx.x & 63) == 0) atomicAdd(&gs_dbg[i], (unsigned long long)(t)); } while (0)
#else
#define GS_STAMP_T() 0ull
#define GS_STAMP_ADD(i, t) do { } while (0)
#endif
#if defined(GS_STRAUS_INLINE)
#define GS_STRAUS GS_HD
#else
#define GS_STRAUS GS_HD_NOINLINE
#endif
template <class C, class F, int TMAX, int W>
GS_STRAUS void jac_straus_build(Aff<F>* at, F& zback, const Aff<F>* ps, int nt, Jac<F>* tab) {
  constexpr int NE = 1 << (W - 1);
  for (int t = 0; t < nt; t++) smul_build_table_n(tab + t * NE, ps[t], NE);
  table_global_z<C>(at, tab, NE * nt, zback);  // ONE isomorphic curve for all the terms' tables
}
// `tabs` (optional): per-term table pointers instead of the lane's own contiguous `at` -- window tables of the BASES
// that many lanes share (k_var_tab: one table per (equation, base), true affine entries, zback = 1); `negm` bit t
// then stands for "-P_t".
// GTAB: the tables (`at` / `tabs`) are GLOBAL memory (the kernels' lane-contiguous workspaces, the shared base tables):
// their look-ups are global_load; false (the single-call helper's local staging) keeps generic loads.
// LDSKB: LDS a wave may take for its digits (36 KB at one wave per SIMD, 18 KB in kernels built for two).
template <class C, class F, int TMAX, int W, bool GTAB = false, int LDSKB = 36>
GS_STRAUS void jac_straus_run(Jac<F>& rout, const Fr<C>* ks, int nt, const Aff<F>* at, const F& zback,
                                   const Aff<F>* const* tabs = nullptr, uint32_t negm = 0) {
  constexpr int NE = 1 << (W - 1);
  Jac<F> r;  // the running sum stays a local (registers): a reference parameter is memory at every step
  if constexpr (C::HAS_ENDO) {
    typedef EndoShape<C, F> E;
    constexpr int ND = E::nd(W);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GS_STRAUS_OLD_LOOP)
    // ---- device main loop (round 4).  At one wave per SIMD every vector-memory wait is exposed in full, and
    // s_waitcnt vmcnt counts in order: ANY load a step waits for also waits for everything requested before it.  So:
    //  * the digits live in LDS (their reads count on lgkmcnt, not vmcnt; rows of an odd number of dwords per lane:
    //    conflict-free), where the private segment cost a scratch_load_sbyte + vmcnt(0) per table look-up;
    //  * the table entry of the NEXT non-trivial step is requested before the current addition starts (one step =
    //    ~6 k instructions: the HBM latency of the lane-contiguous tables disappears behind it) -- which only works
    //    because nothing else in the loop waits on vmcnt: the G1 point operations are register-only subroutines
    //    (jac_dbl_ip / jac_madd_ip) and the running sum never has its address taken.
    unsigned long long st0 = GS_STAMP_T(), st_dbl = 0, st_add = 0;
    constexpr int NSL = E::NS * ND;                       // digits per term
    constexpr int ROWB = ((TMAX * NSL + 3) / 4 * 4);      // bytes per lane, whole dwords ...
    constexpr int ROW = ((ROWB / 4) % 2 == 0) ? ROWB + 4 : ROWB;  // ... an ODD number of them
    constexpr bool LDS_DG = (size_t)ROW * 64 <= (size_t)LDSKB * 1024;  // (else: private frame, as rounds 1-3 had them)
    __shared__ int8_t sdg[LDS_DG ? ROW * 64 : 4];
    int8_t ldg[LDS_DG ? 1 : TMAX * NSL];
    int8_t* const dgp = LDS_DG ? &sdg[(threadIdx.x & 63) * ROW] : ldg;
    uint32_t sgm = 0;  // bit t * NS + s: stream s of term t is negated
    for (int t = 0; t < nt; t++) {
      uint8_t sg1[E::NS];
      endo_digits<C, W>(dgp + t * NSL, sg1, ks[t], (const Jac<F>*)nullptr);
      for (int s2 = 0; s2 < E::NS; s2++) sgm |= (uint32_t)(sg1[s2] != 0) << (t * E::NS + s2);
    }
    auto digit = [&](int i, int k) -> int {  // slot k = t * NS + s of window i
      const int idx = (k / E::NS) * NSL + (k % E::NS) * ND + i;
      if constexpr (LDS_DG) return (int)sdg[(threadIdx.x & 63) * ROW + idx];
      return (int)ldg[idx];
    };
    jac_set_inf(r);
    const int nk = nt * E::NS;
    int top = 0;  // highest window with a non-zero digit: short scalars skip their leading doublings
    for (int k = 0; k < nk; k++)
      for (int i = ND - 1; i > top; i--)
        if (digit(i, k) != 0) top = i;
    auto entry = [&](int i, int k) -> const Aff<F>* {  // (a zero digit requests entry 0: harmless, never used)
      int a = digit(i, k);
      a = a < 0 ? -a : a;
      const int idx = a ? a - 1 : 0, t = k / E::NS;
      return tabs ? tabs[t] + idx : at + t * NE + idx;
    };
    // (limb-wise loads: a struct copy from a pointer is a memcpy into a private-frame temporary that the compiler then
    // copies again, 16 bytes per load + wait + store)
    auto ldaff = [](const Aff<F>* p) -> Aff<F> {
      Aff<F> v;
      constexpr int NWD = (int)(sizeof(Aff<F>) / sizeof(limb_t));
      if constexpr (GTAB) {
        // global memory: global_load, which counts on vmcnt only -- a FLAT load also counts on lgkmcnt, and the next
        // digit read from LDS (s_waitcnt lgkmcnt(0)) would then wait for the whole prefetch
        typedef const __attribute__((address_space(1))) limb_t* gptr;
        gptr w = (gptr) reinterpret_cast<const limb_t*>(p);
#pragma unroll
        for (int q = 0; q < NWD; q++) reinterpret_cast<limb_t*>(&v)[q] = w[q];
      } else {
        const limb_t* w = reinterpret_cast<const limb_t*>(p);
#pragma unroll
        for (int q = 0; q < NWD; q++) reinterpret_cast<limb_t*>(&v)[q] = w[q];
      }
      return v;
    };
    Aff<F> e = ldaff(entry(top, 0));
    unsigned long long st1 = GS_STAMP_T();
    for (int i = top; i >= 0; i--) {
      if (i != top) {
        unsigned long long sa = GS_STAMP_T();
#pragma unroll 1
        for (int d4 = 0; d4 < W; d4++) jac_dbl_ip(r);
        st_dbl += GS_STAMP_T() - sa;
      }
#pragma unroll 1
      for (int k = 0; k < nk; k++) {
        Aff<F> en = e;
        {
          int ni = i, k2 = k + 1;
          if (k2 == nk) ni = i - 1, k2 = 0;
          if (ni >= 0) en = ldaff(entry(ni, k2));  // requested now, consumed after this step's addition
        }
        const int a = digit(i, k);
        unsigned long long sb = GS_STAMP_T();
        if (a != 0) {
          const int t = k / E::NS, s2 = k % E::NS;
          Aff<F> u = e;
          endo_apply<C>(u, s2);
          if (((a < 0) != (((sgm >> k) & 1) != 0)) != (((negm >> t) & 1) != 0)) u.y = neg(u.y);
          jac_madd_ip(r, u);
        }
        st_add += GS_STAMP_T() - sb;
        e = en;
      }
    }
    {
      unsigned long long st2 = GS_STAMP_T();
      GS_STAMP_ADD(0, st1 - st0);
      GS_STAMP_ADD(1, st2 - st1);
      GS_STAMP_ADD(2, st_dbl);
      GS_STAMP_ADD(3, st_add);
      GS_STAMP_ADD(4, 1);
    }
#else
    int8_t dg[TMAX][E::NS * ND];
    uint8_t sg[TMAX][E::NS];
    for (int t = 0; t < nt; t++) endo_digits<C, W>(dg[t], sg[t], ks[t], (const Jac<F>*)nullptr);
    jac_set_inf(r);
    int top = 0;  // highest window with a non-zero digit: short scalars skip their leading doublings
    for (int t = 0; t < nt; t++)
      for (int s = 0; s < E::NS; s++)
        for (int i = ND - 1; i > top; i--)
          if (dg[t][s * ND + i] != 0) top = i;
    for (int i = top; i >= 0; i--) {
      if (i != top) {
#pragma unroll 1
        for (int d4 = 0; d4 < W; d4++) jac_dbl_ip(r);
      }
      for (int t = 0; t < nt; t++)
        for (int s = 0; s < E::NS; s++) {
          int a = dg[t][s * ND + i];
          if (a == 0) continue;
          Aff<F> e = tabs ? tabs[t][(a < 0 ? -a : a) - 1] : at[t * NE + (a < 0 ? -a : a) - 1];
          endo_apply<C>(e, s);
          if (((a < 0) != (sg[t][s] != 0)) != (((negm >> t) & 1) != 0)) e.y = neg(e.y);
          jac_madd_ip(r, e);
        }
    }
#endif
  } else {
    static_assert(C::HAS_ENDO || W == 4, "plain curves: width 4 only");
    constexpr int ND = (FrM<C>::BITS + 3) / 4 + 1;
    int8_t dg[TMAX][ND];
    for (int t = 0; t < nt; t++) recode_w4<FrM<C>>(dg[t], ND, ks[t]);
    jac_set_inf(r);
    int top = 0;
    for (int t = 0; t < nt; t++)
      for (int i = ND - 1; i > top; i--)
        if (dg[t][i] != 0) top = i;
    for (int i = top; i >= 0; i--) {
      if (i != top) {
#pragma unroll 1
        for (int d4 = 0; d4 < 4; d4++) jac_dbl(r, r);
      }
      for (int t = 0; t < nt; t++) {
        int a = dg[t][i];
        if (a == 0) continue;
        Aff<F> e = at[t * NE + (a < 0 ? -a : a) - 1];
        if (a < 0) e.y = neg(e.y);
        jac_madd(r, r, e);
      }
    }
  }
  r.z = mul(r.z, zback);
  rout = r;
}
template <class C, class F, int TMAX>
GS_HD void jac_msm_straus_at(Jac<F>& rout, const Aff<F>* ps, const Fr<C>* ks, int nt, Aff<F>* at) {
  F zback;
  Jac<F> tab[TMAX * 8];  // (4-bit windows, small groups: the single-call helper keeps its staging local)
  jac_straus_build<C, F, TMAX, 4>(at, zback, ps, nt, tab);
  jac_straus_run<C, F, TMAX, 4>(rout, ks, nt, at, zback);
}

}  // namespace gs
