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
template <class F> GS_HD_NOINLINE void jac_dbl(Jac<F>& r, const Jac<F>& p) {
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
template <class F> GS_HD_NOINLINE void jac_madd(Jac<F>& r, const Jac<F>& p, const Aff<F>& q) {
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
template <class F> GS_HD_NOINLINE void jac_add(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
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
  jac_dbl(tab[1], tab[0]);
  jac_madd(tab[2], tab[1], p);
  jac_dbl(tab[3], tab[1]);
  jac_madd(tab[4], tab[3], p);
  jac_dbl(tab[5], tab[2]);
  jac_madd(tab[6], tab[5], p);
  jac_dbl(tab[7], tab[3]);
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
template <class F, class M> GS_HD_NOINLINE void jac_smul(Jac<F>& r, const Aff<F>& p, const Fe<M>& k) {
  constexpr int ND = (M::BITS + 3) / 4 + 1;  // one spare digit for the signed carry
  Jac<F> tab[8];
  int8_t dg[ND];
  smul_build_table(tab, p);
  recode_w4<M>(dg, ND, k);
  jac_set_inf(r);
  for (int i = ND - 1; i >= 0; i--) {
    if (i != ND - 1) {
      jac_dbl(r, r);
      jac_dbl(r, r);
      jac_dbl(r, r);
      jac_dbl(r, r);
    }
    int d = dg[i];
    if (d != 0) {
      int a = d < 0 ? -d : d;
      Jac<F> t = tab[a - 1];
      if (d < 0) t.y = neg(t.y);
      jac_add(r, r, t);
    }
  }
}

}  // namespace gs
