// Optimal-ate pairing pieces: Miller-loop doubling/addition steps on the twist
// in homogeneous projective coordinates, sparse line multiplication, and the
// final exponentiation with ARKWORKS' exponent.
//
// Replaces E::pairing / E::multi_pairing (ark-ec ^0.5 models::bls12 / models::bn,
// external to the reference; call sites src/data_structures.rs:484-502).
// Result-defining convention: for BLS12 the hard part raises to
// (x-1)^2 (x+p)(x^2+p^2-1) + 3 = 3 (p^4-p^2+1)/r  (eprint 2020/875), i.e. the
// cube of the textbook reduced pairing.  Any Miller-loop variant is admissible:
// factors in proper subfields vanish under the final exponentiation and the
// fully exponentiated value is unique.
//
// Line derivation (M-type twist, untwist (x',y') -> (x'/w^2, y'/w^3)): the
// tangent/chord through T with twist-slope L evaluated at P=(xP,yP), scaled by
// w^3 (in Fp4) and an Fp2 factor, is
//     l = ell0 + (ellx * xP) w^2 + (elly * yP) w^3
// with w^2 = v (slot 1) and w^3 = v w (slot 4): f.mul_by_014.  D-type twist:
//     l = (elly * yP) + (ellx * xP) w + ell0 w^3 : f.mul_by_034.
#pragma once
#include "gs_curve.cuh"

namespace gs {

template <class C> struct Proj2 {  // homogeneous projective point on the twist
  Fp2<C> x, y, z;
};
template <class C> struct Line {
  Fp2<C> l0, lx, ly;
};

template <class C> GS_HD Fp2<C> twist_b3() {
  Fp2<C> r;
#pragma unroll
  for (int i = 0; i < C::L; i++) {
    r.c0.v[i] = C::B2X3_28[0][i];
    r.c1.v[i] = C::B2X3_28[1][i];
  }
  return r;
}

// T <- 2T, returns the tangent line coefficients (all N).  The new point is the
// classical (X3, Y3, Z3) scaled by 4 -- the same projective point, no halvings:
//   X3 = 2XY (B - F), Y3 = (B + F)^2 - 12 E^2, Z3 = 4 B H
// with B = Y^2, E = 3 b' Z^2, F = 3E, H = 2YZ.
template <class C> GS_ML void miller_dbl(Proj2<C>& t, Line<C>& l) {
  Fp2<C> xy = mul(t.x, t.y);
  Fp2<C> b = sqr(t.y);
  Fp2<C> c = sqr(t.z);
  Fp2<C> e;                                                   // 3 b' Z^2
  if constexpr (C::B2X3_XI_K != 0) {
    // 3 b' = k xi with k small (BLS12-381: 12): linear operations and one value reduction (|V| < 3 k p before it)
    // instead of an Fp2 multiplication
    static_assert(C::B2X3_XI_K % 4 == 0 && C::B2X3_XI_K <= 12, "limb growth below: 2 x 3, then x 4");
    Fp2<C> x = norm(mul_small(norm(mul_small(mul_xi(c), C::B2X3_XI_K / 4)), 4));
    e.c0 = vreduce(x.c0);
    e.c1 = vreduce(x.c1);
  } else {
    e = mul(twist_b3<C>(), c);
  }
  Fp2<C> f = norm(add(dbl(e), e));                            // 9 b' Z^2
  Fp2<C> h = norm(sub(sub(sqr_l2(add(t.y, t.z)), b), c));     // 2 Y Z
  Fp2<C> j = sqr(t.x);
  Fp2<C> e2 = sqr(e);
  l.l0 = norm(sub(b, e));                                     // Y^2 - 3 b' Z^2
  l.lx = norm(neg(add(dbl(j), j)));                           // -3 X^2
  l.ly = h;                                                   // 2 Y Z
  Fp2<C> e2x4 = norm(dbl(dbl(e2)));
  t.x = mul(norm(dbl(xy)), norm(sub(b, f)));
  t.y = norm(sub(sqr(norm(add(b, f))), add(dbl(e2x4), e2x4)));
  t.z = mul(b, norm(dbl(dbl(h))));
}

// T <- T + Q (Q affine on the twist), returns the chord line coefficients (all N).
template <class C> GS_ML void miller_add(Proj2<C>& t, Line<C>& l, const Aff<Fp2<C>>& q) {
  Fp2<C> theta = norm(sub(t.y, mul(q.y, t.z)));
  Fp2<C> lambda = norm(sub(t.x, mul(q.x, t.z)));
  Fp2<C> c = sqr(theta);
  Fp2<C> d = sqr(lambda);
  Fp2<C> e = mul(lambda, d);
  Fp2<C> f = mul(t.z, c);
  Fp2<C> g = mul(t.x, d);
  Fp2<C> h = norm(sub(add(e, f), dbl(g)));
  l.l0 = norm(sub(mul(theta, q.x), mul(lambda, q.y)));
  l.lx = neg(theta);
  l.ly = lambda;
  t.x = mul(lambda, h);
  t.y = norm(sub(mul(theta, norm(sub(g, h))), mul(e, t.y)));
  t.z = mul(t.z, e);
}

// Lines evaluated at P can go into the accumulator in PAIRS (f12_mul_by_lines2: 23 instead of 26 Fp2 multiplications per
// two lines; a line left over is held until the next one arrives or the accumulator is needed -- before its next squaring /
// at the end -- where it goes in on its own).  Off since round 3 (see pair_lines below); kept for GS_LINES2_ALL.
template <class C> struct LineAcc {
  ELine<C> pend;
  bool has = false;
  GS_HD void add(Fp12<C>& f, const Line<C>& l, const Aff<Fq<C>>& p) {
    Fp2<C> cx = mul_fp(l.lx, p.x), cy = mul_fp(l.ly, p.y);
#if defined(GS_LINES2_ALL)
    constexpr bool pair_lines = true;
#elif defined(GS_LINES2_NONE)
    constexpr bool pair_lines = false;
#else
    // Round 3: never.  With the lane-pair kernel and the dot-product form of the sparse product in its register-friendly
    // call order (gs_tower.cuh) the unpaired form wins on BOTH curves: BN254 k_miller.pairdpp 107.8 -> 100.9 ms at 2^16
    // (`profiles/r3/ab_nolines2_bn254.txt`), BLS12-381 157.4-158.8 ms unpaired against 164.8-165.2 ms paired
    // (`profiles/r3/ab_lines2_bls.txt`).  GS_LINES2_ALL brings the paired form back (round 1-2: it took 8 % off the
    // BN254 twin kernel).
    constexpr bool pair_lines = false;
#endif
    // (round 2, with the dot-product form of the sparse product: BN254 paired 130.4 ms, unpaired 130.9 ms at 2^16)
    if (!pair_lines) {  // measured on gfx950: pairing is a wash on BLS12-381 (the sparse product is the more
      if (C::TWIST_M)  // register-friendly one) and takes 8 % off k_miller on BN254: lines are paired there only
        f12_mul_by_014(f, l.l0, cx, cy);
      else
        f12_mul_by_034(f, cy, cx, l.l0);
      return;
    }
    ELine<C> e;
    if (C::TWIST_M) {
      e.a = l.l0;
      e.b = cx;
      e.c = cy;
    } else {
      e.a = cy;
      e.b = cx;
      e.c = l.l0;
    }
    if (has) {
      f12_mul_by_lines2(f, pend, e);
      has = false;
    } else {
      pend = e;
      has = true;
    }
  }
  GS_HD void flush(Fp12<C>& f) {
    if (has) {
      f12_mul_by_line(f, pend);
      has = false;
    }
  }
};

// Multi-Miller loop over `np` pairs held in memory; pairs with an identity
// argument are skipped (they contribute 1).  Result is NOT exponentiated.
// `ts` is caller-provided scratch for the np running twist points.
// Lines of a FIXED G2 argument (the CRS elements v, W2 of every GS verification) do not depend on the batch: they are
// tabulated once per CRS in consumption order -- per loop digit the tangent line, then the chord line if the digit is
// non-zero, then (BN) the two Frobenius chords -- and a pair whose `fixed[k]` is set reads them instead of stepping its
// own twist point (25 of the 68 Fq multiplications of a pair-step).
template <class C> constexpr int miller_line_count() {
  int n = 0;
  for (int i = C::LOOP_LEN - 2; i >= 0; i--) n += 1 + (C::LOOP[i] != 0 ? 1 : 0);
  return n + (C::IS_BN ? 2 : 0);
}
template <class C> GS_HD_NOINLINE void miller_line_table(Line<C>* out, const Aff<Fp2<C>>& q) {
  Proj2<C> t;
  t.x = q.x;
  t.y = q.y;
  t.z = one_of<Fp2<C>>();
  int n = 0;
  for (int i = C::LOOP_LEN - 2; i >= 0; i--) {
    miller_dbl(t, out[n++]);
    int d = C::LOOP[i];
    if (d != 0) {
      Aff<Fp2<C>> qq = q;
      if (d < 0) qq.y = neg(qq.y);
      miller_add(t, out[n++], qq);
    }
  }
  if (C::IS_BN) {
    Aff<Fp2<C>> q1, q2;
    q1.x = mul(conj(q.x), frob_coeff<C>(1, 2));
    q1.y = mul(conj(q.y), frob_coeff<C>(1, 3));
    q2.x = mul(q.x, frob_coeff<C>(2, 2));
    q2.y = neg(mul(q.y, frob_coeff<C>(2, 3)));
    miller_add(t, out[n++], q1);
    miller_add(t, out[n++], q2);
  }
}

template <class C>
GS_HD_NOINLINE void multi_miller(Fp12<C>& fout, const Aff<Fq<C>>* ps, const Aff<Fp2<C>>* qs, int np, Proj2<C>* ts,
                                 bool* live, const Line<C>* const* fixed = nullptr) {
  // A LOCAL accumulator, written back at the end.  Round 2 measured no difference (109.7 vs 110.5 ms, batched verifier
  // at 2^16); with the register-friendly call order of the sparse product (gs_tower.cuh, round 3) the local form wins:
  // k_miller.rlc 97.8 / 97.9 -> 94.4 / 93.6 ms.
  Fp12<C> f;
  f12_one(f);
  bool any = false;
  for (int k = 0; k < np; k++) {
    live[k] = !(aff_is_inf(ps[k]) || aff_is_inf(qs[k]));
    any |= live[k];
    ts[k].x = qs[k].x;
    ts[k].y = qs[k].y;
    ts[k].z = one_of<Fp2<C>>();
  }
  if (!any) {
    fout = f;
    return;
  }
  Line<C> l;
  LineAcc<C> acc;
  int li = 0;  // position in the line tables of the fixed arguments
  for (int i = C::LOOP_LEN - 2; i >= 0; i--) {
    acc.flush(f);
    f12_sqr(f, f);
    for (int k = 0; k < np; k++) {
      if (!live[k]) continue;
      if (fixed && fixed[k]) {
        acc.add(f, fixed[k][li], ps[k]);  // coefficients straight from the table
        continue;
      }
      miller_dbl(ts[k], l);
      acc.add(f, l, ps[k]);
    }
    li++;
    int d = C::LOOP[i];
    if (d != 0) {
      for (int k = 0; k < np; k++) {
        if (!live[k]) continue;
        if (fixed && fixed[k]) {
          acc.add(f, fixed[k][li], ps[k]);
          continue;
        }
        Aff<Fp2<C>> q = qs[k];
        if (d < 0) q.y = neg(q.y);
        miller_add(ts[k], l, q);
        acc.add(f, l, ps[k]);
      }
      li++;
    }
  }
  if (C::IS_BN) {
    // two extra lines with pi(Q) and -pi^2(Q); twist Frobenius:
    // pi(x', y') = (conj(x') * xi^((p-1)/3), conj(y') * xi^((p-1)/2))
    for (int k = 0; k < np; k++) {
      if (!live[k]) continue;
      if (fixed && fixed[k]) {
        acc.add(f, fixed[k][li], ps[k]);
        acc.add(f, fixed[k][li + 1], ps[k]);
        continue;
      }
      Aff<Fp2<C>> q1, q2;
      q1.x = mul(conj(qs[k].x), frob_coeff<C>(1, 2));
      q1.y = mul(conj(qs[k].y), frob_coeff<C>(1, 3));
      q2.x = mul(qs[k].x, frob_coeff<C>(2, 2));
      q2.y = neg(mul(qs[k].y, frob_coeff<C>(2, 3)));
      miller_add(ts[k], l, q1);
      acc.add(f, l, ps[k]);
      miller_add(ts[k], l, q2);
      acc.add(f, l, ps[k]);
    }
  }
  acc.flush(f);
  if (C::LOOP_NEG) f12_conj(f, f);
  fout = f;
}

// Twin multi-Miller loop: every G2 argument Q_k of the GS verification equation is
// paired with TWO G1 arguments (the a = 0 and a = 1 components of the same
// commitment-group element, data_structures.rs:484-502), so the tangent/chord
// line of Q_k is computed once and evaluated at both: f0 *= l(P0_k), f1 *= l(P1_k).
template <class C>
GS_HD_NOINLINE void multi_miller2(Fp12<C>& f0out, Fp12<C>& f1out, const Aff<Fq<C>>* p0, const Aff<Fq<C>>* p1,
                                  const Aff<Fp2<C>>* qs, int np, Proj2<C>* ts, uint8_t* live,
                                  const Line<C>* const* fixed = nullptr) {
  // (f0 / f1 stay behind the references: as locals the allocator turns their 336 dwords into one-dword spills and the
  // twin kernel goes from 194 to 228 ms at 2^16; the 16-byte loads and stores through the pointers are the cheaper
  // register file extension)
  Fp12<C>&f0 = f0out, &f1 = f1out;  // (one of the two as a local: 178.5 vs 176.4 ms)
  f12_one(f0);
  f12_one(f1);
  bool any = false;
  for (int k = 0; k < np; k++) {
    bool q_ok = !aff_is_inf(qs[k]);
    uint8_t l = 0;
    if (q_ok && !aff_is_inf(p0[k])) l |= 1;
    if (q_ok && !aff_is_inf(p1[k])) l |= 2;
    live[k] = l;
    any |= (l != 0);
    ts[k].x = qs[k].x;
    ts[k].y = qs[k].y;
    ts[k].z = one_of<Fp2<C>>();
  }
  if (!any) return;
  Line<C> l;
  LineAcc<C> acc0, acc1;
  int li = 0;
  for (int i = C::LOOP_LEN - 2; i >= 0; i--) {
    acc0.flush(f0);
    acc1.flush(f1);
    f12_sqr(f0, f0);
    f12_sqr(f1, f1);
    for (int k = 0; k < np; k++) {
      if (!live[k]) continue;
      const Line<C>* lp = &l;
      if (fixed && fixed[k])
        lp = &fixed[k][li];
      else
        miller_dbl(ts[k], l);
      if (live[k] & 1) acc0.add(f0, *lp, p0[k]);
      if (live[k] & 2) acc1.add(f1, *lp, p1[k]);
    }
    li++;
    int d = C::LOOP[i];
    if (d != 0) {
      for (int k = 0; k < np; k++) {
        if (!live[k]) continue;
        const Line<C>* lp = &l;
        if (fixed && fixed[k]) {
          lp = &fixed[k][li];
        } else {
          Aff<Fp2<C>> q = qs[k];
          if (d < 0) q.y = neg(q.y);
          miller_add(ts[k], l, q);
        }
        if (live[k] & 1) acc0.add(f0, *lp, p0[k]);
        if (live[k] & 2) acc1.add(f1, *lp, p1[k]);
      }
      li++;
    }
  }
  if (C::IS_BN) {
    for (int k = 0; k < np; k++) {
      if (!live[k]) continue;
      if (fixed && fixed[k]) {
        for (int e = 0; e < 2; e++) {
          if (live[k] & 1) acc0.add(f0, fixed[k][li + e], p0[k]);
          if (live[k] & 2) acc1.add(f1, fixed[k][li + e], p1[k]);
        }
        continue;
      }
      Aff<Fp2<C>> q1, q2;
      q1.x = mul(conj(qs[k].x), frob_coeff<C>(1, 2));
      q1.y = mul(conj(qs[k].y), frob_coeff<C>(1, 3));
      q2.x = mul(qs[k].x, frob_coeff<C>(2, 2));
      q2.y = neg(mul(qs[k].y, frob_coeff<C>(2, 3)));
      miller_add(ts[k], l, q1);
      if (live[k] & 1) acc0.add(f0, l, p0[k]);
      if (live[k] & 2) acc1.add(f1, l, p1[k]);
      miller_add(ts[k], l, q2);
      if (live[k] & 1) acc0.add(f0, l, p0[k]);
      if (live[k] & 2) acc1.add(f1, l, p1[k]);
    }
  }
  acc0.flush(f0);
  acc1.flush(f1);
  if (C::LOOP_NEG) {
    f12_conj(f0, f0);
    f12_conj(f1, f1);
  }
}

// Pair-cooperative twin Miller loop: the twin loop above spread over TWO lanes.  Lanes 2i and 2i + 1 of a wave work on
// the same (equation, task): lane a owns the accumulator of component a for the whole loop (ONE accumulator per lane:
// it fits the register file, the two of multi_miller2 do not and stream through memory at every line product), the
// stepping twist points of the task are dealt out alternately (lane a steps triples a, a + 2, ...), and every tangent /
// chord line crosses to the partner lane once through the exchange policy X (LDS slots or DPP on the device).  Total
// work is that of the twin loop: the same squarings per accumulator, every line computed once and evaluated at both
// G1 components (data_structures.rs:494-502).
//   ps[k]     the lane's OWN G1 component of every triple, k < np (stepping triples first: k < nstep)
//   qown[r]   the twist point of the lane's r-th stepping triple, k = 2 r + a, r < (nstep + 1) / 2 (with nstep odd the
//             last lane-1 entry is any valid point: it is stepped and never consumed)
//   qok       bit k: Q_k is not the identity (the same on both lanes)
//   fixed[k]  line table of a CRS G2 argument (k >= nstep), consumed as in multi_miller
// X::put(l) hands the lane's line to the exchange, X::get() returns the partner's; both lanes of a pair always reach
// both together.  Two phases so that a policy that parks the line in memory (LDS) can fetch the partner's line AFTER the
// lane's own line product: held in registers across that product (live set ~440 dwords) it would be spilled and
// reloaded through the private segment, 84 dwords each way per step.
template <class C, class X>
GS_HD void multi_miller_pair(Fp12<C>& fout, int a, const Aff<Fq<C>>* ps, const Aff<Fp2<C>>* qown, uint32_t qok,
                                      int nstep, int np, Proj2<C>* ts, const Line<C>* const* fixed, X& xch) {
  Fp12<C> f;  // local: the whole point of this shape
  f12_one(f);
  uint32_t live = 0;
  for (int k = 0; k < np; k++)
    if (((qok >> k) & 1) && !aff_is_inf(ps[k])) live |= 1u << k;
  const int rounds = (nstep + 1) / 2;
  for (int r = 0; r < rounds; r++) {
    const Aff<Fp2<C>>& q = qown[r];
    ts[r].x = q.x;
    ts[r].y = q.y;
    ts[r].z = one_of<Fp2<C>>();
  }
  Line<C> l, lp;
  LineAcc<C> acc;
  int li = 0;
  // The products of one round: the lane's own line at its own P, the partner's line at the lane's P for the partner's
  // triple.  Both lanes of a pair sit in the same wave, so what the WAVE executes is the union of its lanes' branches.
  // A full round is two products on every lane.  The last round of an ODD number of stepping triples has one line only:
  // lane 0 owes its own line, lane 1 the partner's -- taken as two branches the wave would run two products of which
  // each lane uses one (until round 3 it did: a lane of 1 stepping triple cost as much as one of 2, 9.2 ms instead of
  // ~6.5 ms for the 20-task plans of small batches).  So: one product slot whose operands are selected per lane.
  auto products = [&](const Line<C>& mine, int ko, int kp) {
    const bool own_ok = ko < nstep && ((live >> ko) & 1), par_ok = kp < nstep && ((live >> kp) & 1);
    xch.put(mine);
    if (own_ok && par_ok) {
      acc.add(f, mine, ps[ko]);
      lp = xch.get();
      acc.add(f, lp, ps[kp]);
    } else {
      lp = xch.get();
      if (own_ok || par_ok) {
        Line<C> use;  // (word-wise selects, not a branch per operand: a branch would bring the two products back)
        use.l0 = select(own_ok, mine.l0, lp.l0);
        use.lx = select(own_ok, mine.lx, lp.lx);
        use.ly = select(own_ok, mine.ly, lp.ly);
        acc.add(f, use, ps[own_ok ? ko : kp]);
      }
    }
  };
  for (int i = C::LOOP_LEN - 2; i >= 0; i--) {
    acc.flush(f);
    f12_sqr(f, f);
    for (int r = 0; r < rounds; r++) {
      const int ko = 2 * r + a, kp = 2 * r + 1 - a;
      miller_dbl(ts[r], l);
      products(l, ko, kp);
    }
    for (int k = nstep; k < np; k++)
      if ((live >> k) & 1) acc.add(f, fixed[k][li], ps[k]);
    li++;
    int d = C::LOOP[i];
    if (d != 0) {
      for (int r = 0; r < rounds; r++) {
        const int ko = 2 * r + a, kp = 2 * r + 1 - a;
        Aff<Fp2<C>> q = qown[r];
        if (d < 0) q.y = neg(q.y);
        miller_add(ts[r], l, q);
        products(l, ko, kp);
      }
      for (int k = nstep; k < np; k++)
        if ((live >> k) & 1) acc.add(f, fixed[k][li], ps[k]);
      li++;
    }
  }
  if (C::IS_BN) {
    for (int e = 0; e < 2; e++) {
      for (int r = 0; r < rounds; r++) {
        const int ko = 2 * r + a, kp = 2 * r + 1 - a;
        const Aff<Fp2<C>>& q = qown[r];
        Aff<Fp2<C>> qf;
        if (e == 0) {
          qf.x = mul(conj(q.x), frob_coeff<C>(1, 2));
          qf.y = mul(conj(q.y), frob_coeff<C>(1, 3));
        } else {
          qf.x = mul(q.x, frob_coeff<C>(2, 2));
          qf.y = neg(mul(q.y, frob_coeff<C>(2, 3)));
        }
        miller_add(ts[r], l, qf);
        products(l, ko, kp);
      }
      for (int k = nstep; k < np; k++)
        if ((live >> k) & 1) acc.add(f, fixed[k][li + e], ps[k]);
    }
  }
  acc.flush(f);
  if (C::LOOP_NEG) f12_conj(f, f);
  fout = f;
}

// f^|x| by square-and-multiply over the 64-bit curve parameter, cyclotomic
// squarings; then conjugate if x < 0 (so the result is f^x).
template <class C> GS_HD_NOINLINE void f12_exp_by_x_gs(Fp12<C>& r, const Fp12<C>& f) {
  Fp12<C> acc = f;
  int top = 63;
  while (!((C::X_ABS >> top) & 1)) top--;
  int since = 0;
  for (int i = top - 1; i >= 0; i--) {
#if defined(GS_FE_INLINE)
    f12_cyclo_sqr_inl(acc, acc);
#else
    f12_cyclo_sqr(acc, acc);
#endif
    // Granger-Scott squaring feeds 2*z back linearly: values double per step; a
    // full multiplication contracts them again, otherwise reduce every 3rd step
    if ((C::X_ABS >> i) & 1) {
      // `acc` must not have its address taken anywhere in this loop: it then lives in registers (AGPRs) from one
      // squaring to the next instead of crossing memory twice per step (k_final 54.9 -> 44.5 ms at 2^16).  (The product
      // INLINE here, as in f12_exp_by_x_karabina, was measured in round 4 and LOST on BN254 -- k_final 28.1 -> 31.9 ms:
      // with the product's 18 multiplier calls in the loop body the accumulator no longer stays in registers across the
      // squarings, which costs more than the register saves of 26 calls.)
      Fp12<C> t = acc;
      f12_mul(t, t, f);
      acc = t;
      since = 0;
    } else if (++since == 3) {
      f12_vreduce_inl(acc);
      since = 0;
    }
  }
  if (C::X_NEG) f12_conj(acc, acc);
  r = acc;
}

// ---- Karabina's compressed squarings (round 4) ------------------------------------------------------
// In the cyclotomic subgroup the Granger-Scott squaring is three Fp4 squarings on the coefficient pairs
//   (z0, z1) = (c0.c0, c1.c1),  (z2, z3) = (c1.c0, c0.c2),  (z4, z5) = (c0.c1, c1.c2)      [f12_cyclo_sqr_inl]
// and the new (z2, z3, z4, z5) depend on the old (z2, z3, z4, z5) ALONE: a run of squarings can drop (z0, z1) -- two Fp4
// squarings per step instead of three, and a working set of 4 Fp2 that never leaves the registers -- and recover them
// where a value is needed (Karabina, "Squaring in cyclotomic subgroups", Math. Comp. 2013; g_i = z_i):
//     z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2),      z0 = xi (2 z1^2 + z2 z5 - 3 z3 z4) + 1.
// f^|x| = product over the set bits b of x of f^(2^b): ONE chain of top(x) compressed squarings, the (<= 6) values at the
// set bits saved, all of them decompressed behind ONE inversion (Montgomery's trick over their 4 z2; the inversion itself
// is the safegcd of gs_fq28.cuh) and multiplied.  Checked against the big-integer oracle on both towers
// (tools: the closure and the two formulas hold with xi the Fp6 non-residue and w^2 = v).  BLS12-381: x has 6 set
// bits; 63 steps of 4 Fp2 products instead of 6, +6 decompressions (3 squarings + 3 products each) + one shared
// inversion: ~0.85 of the Granger-Scott run's multiply-adds and none of its accumulator traffic.  A z2 that is 0 (f = 1
// -- an all-identity pairing product -- is the case that occurs) sends the lane to the uncompressed routine.  BN254's
// x has 26 set bits: the decompressions would cost more than the squarings save, it keeps the uncompressed run.
template <class C> GS_HD void cyclo_sqr_compressed(Fp2<C>& g2, Fp2<C>& g3, Fp2<C>& g4, Fp2<C>& g5) {
  Fp2<C> t2, t3, t4, t5;
  fp4_sqr(t2, t3, g2, g3);
  fp4_sqr(t4, t5, g4, g5);
  // z2 = 3 xi t5 + 2 z2 ; z3 = 3 t4 - 2 z3 ; z4 = 3 t2 - 2 z4 ; z5 = 3 t3 + 2 z5
  Fp2<C> x5 = norm(mul_xi(t5));
  Fp2<C> z = add(x5, g2);
  g2 = norm(add(dbl(z), x5));
  z = sub(t4, g3);
  g3 = norm(add(dbl(z), t4));
  z = sub(t2, g4);
  g4 = norm(add(dbl(z), t2));
  z = add(t3, g5);
  g5 = norm(add(dbl(z), t3));
}
template <class C> constexpr int x_popcount() {
  int n = 0;
  for (int i = 0; i < 64; i++) n += (int)((C::X_ABS >> i) & 1);
  return n;
}
template <class C> GS_HD_NOINLINE void f12_exp_by_x_karabina(Fp12<C>& r, const Fp12<C>& f) {
  constexpr int NS = x_popcount<C>() - (int)(C::X_ABS & 1);  // values saved along the chain (bit 0 is f itself)
  Fp2<C> sv[NS][4];
  Fp2<C> g2 = f.c1.c0, g3 = f.c0.c2, g4 = f.c0.c1, g5 = f.c1.c2;
  int top = 63;
  while (!((C::X_ABS >> top) & 1)) top--;
  int ns = 0, since = 0;
  unsigned long long sk0 = GS_STAMP_T();
#pragma unroll 1
  for (int i = 1; i <= top; i++) {
    cyclo_sqr_compressed<C>(g2, g3, g4, g5);
    if (++since == 3) {  // the squaring feeds 2 z back linearly: bring the VALUES back to ~[-p, p] now and then
#define GS_VR2(x) x.c0 = vreduce(x.c0), x.c1 = vreduce(x.c1)
      GS_VR2(g2);
      GS_VR2(g3);
      GS_VR2(g4);
      GS_VR2(g5);
#undef GS_VR2
      since = 0;
    }
    if ((C::X_ABS >> i) & 1) {
      sv[ns][0] = g2, sv[ns][1] = g3, sv[ns][2] = g4, sv[ns][3] = g5;
      ns++;
    }
  }
  // 1 / (4 z2) for all saved values behind one inversion
  unsigned long long sk1 = GS_STAMP_T();
  GS_STAMP_ADD(10, sk1 - sk0);  // diagnosis build: the chain of compressed squarings
  Fp2<C> den[NS], pre[NS];
  Fp2<C> acc = one_of<Fp2<C>>();
  bool degenerate = false;
  for (int k = 0; k < NS; k++) {
    den[k] = norm(dbl(dbl(sv[k][0])));
    degenerate |= is_zero(den[k]);
    pre[k] = acc;
    acc = mul(acc, den[k]);
  }
  if (degenerate) {  // z2 = 0 somewhere (f = 1): the uncompressed run handles every element of the subgroup
    f12_exp_by_x_gs(r, f);
    return;
  }
  Fp2<C> suf = inv(acc);
  GS_STAMP_ADD(11, GS_STAMP_T() - sk1);  // prefix products + the shared inversion
  Fp12<C> out;
  bool have = false;
  if (C::X_ABS & 1) {
    out = f;
    have = true;
  }
  for (int k = NS - 1; k >= 0; k--) {
    Fp2<C> i4z2 = mul(suf, pre[k]);
    suf = mul(suf, den[k]);
    const Fp2<C>&z2 = sv[k][0], &z3 = sv[k][1], &z4 = sv[k][2], &z5 = sv[k][3];
    // z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2)        (lazy sum: 2 + 3 + 2 limb-growth units, one carry round)
    Fp2<C> s4 = sqr(z4);
    Fp2<C> num = norm(sub(add(mul_xi(sqr(z5)), add(dbl(s4), s4)), dbl(z3)));
    Fp2<C> z1 = mul(num, i4z2);
    // z0 = xi (2 z1^2 + z2 z5 - 3 z3 z4) + 1
    Fp2<C> b = mul(z3, z4);
    Fp2<C> t = norm(sub(add(dbl(sqr(z1)), mul(z2, z5)), add(dbl(b), b)));
    Fp2<C> z0 = norm(add(mul_xi(t), one_of<Fp2<C>>()));
    Fp12<C> v;
    v.c0.c0 = z0, v.c1.c1 = z1, v.c1.c0 = z2, v.c0.c2 = z3, v.c0.c1 = z4, v.c1.c2 = z5;
    if (have) {
      // INLINE at this one site: every call of the out-of-line f12_mul saves and restores the ~420 callee-saved registers
      // the product clobbers, a dword at a time -- a third of k_final's private-segment traffic, and the product phases
      // of this kernel run at the HBM rate of that traffic (PMC: 16 % of its wave cycles in s_waitcnt).  Inline the
      // product spills only what is live: k_final 36.5 -> 34.1 ms (profiles/r4/ab_fe_inline_*.json).  The same for the
      // 11 products of final_exp_with needs them at ONE site too -- a table-driven loop over a register file of Fp12
      // values, built and measured: 34.8 ms, the dynamic indexing costs more than the 11 register saves (not kept).
      f12_mul_inl(out, out, v);
    } else {
      out = v;
      have = true;
    }
  }
  if (C::X_NEG) f12_conj(out, out);
  r = out;
  GS_STAMP_ADD(12, GS_STAMP_T() - sk0);  // the whole x-power
}
template <class C> GS_HD void f12_exp_by_x(Fp12<C>& r, const Fp12<C>& f) {
#if !defined(GS_NO_KARABINA)
  if constexpr (x_popcount<C>() <= 8) {
    f12_exp_by_x_karabina(r, f);
    return;
  }
#endif
  f12_exp_by_x_gs(r, f);
}

// Final exponentiation, arkworks' exponent.  `ex(r, f)` computes r = f^x: one lane
// (f12_exp_by_x) or a 3-lane group (gs_coop.cuh).
template <class C> struct ExpXLane {
  GS_HD void operator()(Fp12<C>& r, const Fp12<C>& f) const { f12_exp_by_x(r, f); }
};
template <class C, class EX> GS_HD_NOINLINE void final_exp_with(Fp12<C>& out, const Fp12<C>& f, EX& ex) {
  Fp12<C> r, t, y0, y1, y2;
  unsigned long long se0 = GS_STAMP_T();
  if (!C::IS_BN) {
    // The nine general products of the BLS12 exponentiation run at ONE inlined site: the statements between two products
    // are the cases of a switch, each case names the product that follows it (see f12_exp_by_x_karabina for why; measured
    // in three alternations on one box, profiles/r4/ab_fe_switch.txt: step 272.95 -> 271.7 ms, k_final 34.0 -> 33.8 ms).
#pragma unroll 1
    for (int step = 0; step < 9; step++) {
      Fp12<C>*pd, *pb;
      const Fp12<C>* pa;
      switch (step) {
        case 0:   // easy part: f^((p^6-1)(p^2+1))
          f12_inv(t, f);
          f12_conj(r, f);
          pd = &r, pa = &r, pb = &t;
          break;
        case 1:
          f12_frob(t, r, 2);
          pd = &r, pa = &r, pb = &t;
          break;
        case 2:   // hard part, eprint 2020/875 (Hayashida-Hayasaka-Teruya): exponent (x-1)^2 (x+p) (x^2+p^2-1) + 3
          GS_STAMP_ADD(13, GS_STAMP_T() - se0);  // easy part
          f12_cyclo_sqr(y0, r);  // r^2
          ex(y1, r);             // r^x
          f12_conj(y2, r);       // r^-1
          pd = &y1, pa = &y1, pb = &y2;  // r^(x-1)
          break;
        case 3:
          ex(y2, y1);            // r^(x(x-1))
          f12_conj(y1, y1);      // r^-(x-1)
          pd = &y1, pa = &y1, pb = &y2;  // r^((x-1)^2)
          break;
        case 4:
          ex(y2, y1);            // ^x
          f12_frob(y1, y1, 1);   // ^p
          pd = &y1, pa = &y1, pb = &y2;  // r^((x-1)^2 (x+p))
          break;
        case 5:
          pd = &r, pa = &r, pb = &y0;    // r^3
          break;
        case 6:
          ex(y0, y1);            // ^x
          ex(y2, y0);            // ^x^2
          f12_frob(y0, y1, 2);   // ^p^2
          f12_conj(y1, y1);      // ^-1
          pd = &y1, pa = &y1, pb = &y2;
          break;
        case 7:
          pd = &y1, pa = &y1, pb = &y0;  // ^(x^2 + p^2 - 1)
          break;
        default:
          pd = &out, pa = &r, pb = &y1;
          break;
      }
      f12_mul_inl(*pd, *pa, *pb);
    }
  } else {
    f12_inv(t, f);
    f12_conj(r, f);
    f12_mul(r, r, t);
    f12_frob(t, r, 2);
    f12_mul(r, r, t);
    GS_STAMP_ADD(13, GS_STAMP_T() - se0);  // easy part
    // BN hard part (Fuentes-Castaneda et al.), as ark-ec models::bn [ark-mem].
    // exp_by_neg_x(f) = f^(-x)
    Fp12<C> y3, y4, y5, y6, y7, y8, y9;
    ex(y0, r);
    f12_conj(y0, y0);           // r^-x
    f12_cyclo_sqr(y1, y0);
    f12_cyclo_sqr(y2, y1);
    f12_mul(y3, y2, y1);
    ex(y4, y3);
    f12_conj(y4, y4);
    f12_cyclo_sqr(y5, y4);
    ex(y6, y5);
    f12_conj(y6, y6);
    f12_conj(y3, y3);
    f12_conj(y6, y6);
    f12_mul(y7, y6, y4);
    f12_mul(y8, y7, y3);
    f12_mul(y9, y8, y1);
    f12_mul(t, y8, y4);         // y10
    f12_mul(t, t, r);           // y11
    f12_frob(y2, y9, 1);        // y12
    f12_mul(y2, y2, t);         // y13
    f12_frob(y8, y8, 2);
    f12_mul(y2, y8, y2);        // y14
    f12_conj(r, r);
    f12_mul(y9, r, y9);         // y15
    f12_frob(y9, y9, 3);
    f12_mul(out, y9, y2);       // y16
  }
}
template <class C> GS_HD void final_exp(Fp12<C>& out, const Fp12<C>& f) {
  ExpXLane<C> ex;
  final_exp_with(out, f, ex);
}

}  // namespace gs
