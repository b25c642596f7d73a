// Canonical wire format of the values that cross the reference's (de)serialisation
// derives (ark-serialize ^0.5: src/data_structures.rs:128,132; src/prover/commit.rs:18,24;
// src/prover/prove.rs:55; src/statement.rs:61-97,117-179; src/generator.rs:35).  This
// file converts ARRAYS of elements between the boundary form of include/gs_amd.h
// (Montgomery limbs) and the byte strings; the struct framing (Vec length prefixes,
// field order) is host-side (groth_sahai_rs_amd/wire.py, include/gs_amd.hpp).
//
// Encodings ([ark-mem]: restated from the published formats, no reference vector exists in-tree):
//   Fr, Fq   canonical integer, little-endian, 32 / 48 bytes
//   GT       12 Fq in the order c0.c0.c0 .. c1.c2.c1
//   BLS12-381 points (ark-bls12-381's zcash-compatible encoding): big-endian; flags in the top bits
//     of the first byte: 0x80 compressed, 0x40 infinity, 0x20 y lexicographically largest (compressed
//     only).  G1: x [|| y]; G2: x.c1 || x.c0 [|| y.c1 || y.c0].
//   BN254 points (ark-ec's default short-Weierstrass encoding): little-endian; flags in the top bits
//     of the LAST byte of the last coordinate written: 0x80 y > -y, 0x40 infinity.
//     G1: x [|| y]; G2: x.c0 || x.c1 [|| y.c0 || y.c1].
//   "y > -y": for Fq the canonical integer exceeds (p-1)/2; for Fp2 decided by c1, by c0 when c1 = 0.
// Decoding checks: canonical coordinates (< p), flag consistency, a square root exists (compressed),
// on the curve (uncompressed), and with `validate` the r-torsion check ark-serialize's Validate::Yes
// performs ([r]P = O; GT: f^r = 1).
#pragma once
#include "gs_pairing.cuh"

namespace gs {

// ---- canonical words <-> internal form -------------------------------------------
template <class C> GS_HD Fq28<C> fq_from_canonical(const uint32_t* w) {
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
  for (int i = 0; i < C::L; i++) k.v[i] = C::RR28[i];
  return mul(t, k);
}
template <class C> GS_HD void fq_to_canonical(uint32_t* w, const Fq28<C>& a) {
  Fq28<C> k = fq_zero<C>();
  k.v[0] = 1;
  Fq28<C> t = norm_full(mul(norm(a), k));  // value in (-p/2, 3p/2)
  Fq28<C> p;
#pragma unroll
  for (int i = 0; i < C::L; i++) p.v[i] = C::P28[i];
  Fq28<C> plus = norm_full(add(t, p)), minus = norm_full(sub(t, p));
  bool negv = t.v[C::L - 1] < 0;
  bool big = minus.v[C::L - 1] >= 0;
  Fq28<C> c = negv ? plus : (big ? minus : t);
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
// a > b on N-word little-endian integers
template <int N> GS_HD bool words_gt(const uint32_t* a, const uint32_t* b) {
  bool gt = false, decided = false;
  for (int i = N - 1; i >= 0; i--) {
    if (!decided && a[i] != b[i]) {
      gt = a[i] > b[i];
      decided = true;
    }
  }
  return gt;
}
template <int N> GS_HD bool words_zero(const uint32_t* a) {
  uint32_t o = 0;
  for (int i = 0; i < N; i++) o |= a[i];
  return o == 0;
}
template <class C> GS_HD bool words_lt_p(const uint32_t* a) {
  uint32_t p[C::N];
  for (int i = 0; i < C::N; i++) p[i] = C::P_WORDS[i];
  return words_gt<C::N>(p, a);
}

// a^e for a constant exponent of C::N words, 4-bit fixed window (as inv28_raw)
template <class C> GS_HD_NOINLINE void fq_pow_words(Fq28<C>& out, const Fq28<C>& a, const uint32_t* e) {
  Fq28<C> tab[16];
  tab[0] = fq_one<C>();
  tab[1] = norm(a);
  for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], tab[1]);
  Fq28<C> r = fq_one<C>();
  bool started = false;
  for (int w = C::N * 8 - 1; w >= 0; w--) {
    uint32_t dgt = (e[w >> 3] >> ((w & 7) * 4)) & 15u;
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
  out = r;
}
// square root for p = 3 (mod 4); ok = false when a is not a square
template <class C> GS_HD Fq28<C> fq_sqrt(const Fq28<C>& a, bool& ok) {
  uint32_t e[C::N];
  for (int i = 0; i < C::N; i++) e[i] = C::SQRT_EXP[i];
  Fq28<C> r;
  fq_pow_words<C>(r, a, e);
  ok = eq(sqr(r), a);
  return r;
}
// square root in Fp2 = Fp[u]/(u^2+1), p = 3 (mod 4): with n = sqrt(a0^2 + a1^2), delta = (a0 +- n)/2,
// c0 = sqrt(delta), c1 = a1 / (2 c0); pure-real / pure-imaginary roots when a1 = 0.
template <class C> GS_HD_NOINLINE Fp2<C> fp2_sqrt(const Fp2<C>& a, bool& ok) {
  Fq28<C> a0 = norm(a.c0), a1 = norm(a.c1);
  Fp2<C> r = {fq_zero<C>(), fq_zero<C>()};
  if (is_zero(a1)) {
    bool sq;
    Fq28<C> s = fq_sqrt<C>(a0, sq);
    if (sq) {
      r.c0 = s;
    } else {
      s = fq_sqrt<C>(neg(a0), sq);  // a0 = -(s^2) = (s u)^2
      r.c1 = s;
    }
    ok = sq;
    return r;
  }
  bool sq;
  Fq28<C> n = fq_sqrt<C>(norm(add(sqr(a0), sqr(a1))), sq);
  if (!sq) {
    ok = false;
    return r;
  }
  Fq28<C> two = norm(dbl(fq_one<C>())), half = inv(two);
  Fq28<C> d = mul(norm(add(a0, n)), half);
  Fq28<C> c0 = fq_sqrt<C>(d, sq);
  if (!sq) {
    d = mul(norm(sub(a0, n)), half);
    c0 = fq_sqrt<C>(d, sq);
  }
  if (!sq) {
    ok = false;
    return r;
  }
  Fq28<C> c1 = mul(a1, inv(norm(dbl(c0))));
  r.c0 = c0;
  r.c1 = c1;
  Fp2<C> chk = sqr(r);
  ok = eq(chk, Fp2<C>{a0, a1});
  return r;
}
template <class C> GS_HD Fq28<C> f_sqrt(const Fq28<C>& a, bool& ok) { return fq_sqrt<C>(a, ok); }
template <class C> GS_HD Fp2<C> f_sqrt(const Fp2<C>& a, bool& ok) { return fp2_sqrt<C>(a, ok); }

// ---- per-field coordinate I/O ------------------------------------------------------
// coordinate = NC canonical Fq values (1 for Fq, 2 for Fp2: c0, c1)
template <class C> GS_HD void coord_words(uint32_t w[][C::N], const Fq28<C>& a) { fq_to_canonical<C>(w[0], a); }
template <class C> GS_HD void coord_words(uint32_t w[][C::N], const Fp2<C>& a) {
  fq_to_canonical<C>(w[0], a.c0);
  fq_to_canonical<C>(w[1], a.c1);
}
template <class C> GS_HD void coord_from_words(Fq28<C>& a, const uint32_t w[][C::N]) { a = fq_from_canonical<C>(w[0]); }
template <class C> GS_HD void coord_from_words(Fp2<C>& a, const uint32_t w[][C::N]) {
  a.c0 = fq_from_canonical<C>(w[0]);
  a.c1 = fq_from_canonical<C>(w[1]);
}
template <class F> struct NCoord;
template <class C> struct NCoord<Fq28<C>> { static constexpr int V = 1; };
template <class C> struct NCoord<Fp2<C>> { static constexpr int V = 2; };

// y > -y on canonical words (NC = 1: Fq; NC = 2: Fp2, c1 decides unless zero)
template <class C, int NC> GS_HD bool y_is_largest(const uint32_t w[][C::N]) {
  uint32_t h[C::N];
  for (int i = 0; i < C::N; i++) h[i] = C::HALF_P[i];
  if (NC == 2 && !words_zero<C::N>(w[1])) return words_gt<C::N>(w[1], h);
  return words_gt<C::N>(w[0], h);
}

constexpr int FQB(int n_words) { return n_words * 4; }
// one canonical Fq <-> FQ bytes.  BLS12-381: big-endian; BN254: little-endian
template <class C> GS_HD void fq_put_bytes(uint8_t* o, const uint32_t* w) {
  constexpr int B = C::N * 4;
  for (int i = 0; i < B; i++) {
    uint8_t b = (uint8_t)(w[i >> 2] >> ((i & 3) * 8));
    if (C::IS_BN) o[i] = b; else o[B - 1 - i] = b;
  }
}
template <class C> GS_HD void fq_get_bytes(uint32_t* w, const uint8_t* in, uint8_t clear_mask, int flag_byte) {
  constexpr int B = C::N * 4;
  for (int i = 0; i < C::N; i++) w[i] = 0;
  for (int i = 0; i < B; i++) {
    int src = C::IS_BN ? i : B - 1 - i;
    uint8_t b = in[src];
    if (src == flag_byte) b &= (uint8_t)~clear_mask;
    w[i >> 2] |= (uint32_t)b << ((i & 3) * 8);
  }
}

// curve equation right-hand side x^3 + b
template <class C> GS_HD Fq28<C> curve_rhs(const Fq28<C>& x) {
  Fq28<C> b;
  for (int i = 0; i < C::L; i++) b.v[i] = C::B1_28[i];
  return norm(add(mul(sqr(x), x), b));
}
template <class C> GS_HD Fp2<C> curve_rhs(const Fp2<C>& x) {
  Fp2<C> b;
  for (int i = 0; i < C::L; i++) {
    b.c0.v[i] = C::B2_28[0][i];
    b.c1.v[i] = C::B2_28[1][i];
  }
  return norm(add(mul(sqr(x), x), b));
}

// ---- prime-order subgroup membership of a curve point (ark-serialize's Validate::Yes) -----------------------------
// Any correct test gives the same verdict.  BLS12-381 uses the endomorphism tests arkworks itself uses (Scott, eprint
// 2021/1130): G1: phi'(P) = -[x^2] P with phi'(x, y) = (beta^2 x, y);  G2: psi(Q) = [x] Q -- 127- and 64-bit
// multiplications instead of 255-bit ones.  BN254: G1 has cofactor 1 (every curve point is in the group); G2 keeps the
// definition [r] Q = O.
template <class F> GS_HD_NOINLINE void jac_mul_bits(Jac<F>& r, const Aff<F>& p, const uint32_t* k, int nbits) {
  jac_set_inf(r);
  for (int i = nbits - 1; i >= 0; i--) {
    jac_dbl(r, r);
    if ((k[i >> 5] >> (i & 31)) & 1) jac_madd(r, r, p);
  }
}
// J == A for a Jacobian J and an affine, finite A
template <class F> GS_HD bool jac_eq_aff(const Jac<F>& j, const Aff<F>& a) {
  if (is_zero(j.z)) return false;
  F z2 = sqr(j.z);
  return eq(mul(a.x, z2), j.x) && eq(mul(a.y, mul(z2, j.z)), j.y);
}
template <class C> GS_HD_NOINLINE bool in_prime_subgroup(const Aff<Fq28<C>>& p) {
  if constexpr (C::IS_BN) {
    return true;  // cofactor 1
  } else {
    uint32_t x2[4];  // x^2 = lambda + 1
    uint64_t cy = 1;
    for (int i = 0; i < 4; i++) {
      uint64_t t = (uint64_t)C::LAMBDA[i] + cy;
      x2[i] = (uint32_t)t;
      cy = t >> 32;
    }
    Jac<Fq28<C>> j;
    jac_mul_bits(j, p, x2, 128);
    Fq28<C> beta;
    for (int i = 0; i < C::L; i++) beta.v[i] = C::BETA_28[i];
    Aff<Fq28<C>> e = {mul(p.x, sqr(beta)), neg(p.y)};  // -phi'(P)
    return jac_eq_aff(j, e);
  }
}
template <class C> GS_HD_NOINLINE bool in_prime_subgroup(const Aff<Fp2<C>>& q) {
  Jac<Fp2<C>> j;
  if constexpr (C::IS_BN) {
    uint32_t r[FrM<C>::N];
    for (int i = 0; i < FrM<C>::N; i++) r[i] = C::R_WORDS[i];
    jac_mul_bits(j, q, r, FrM<C>::BITS);
    return is_zero(j.z);
  } else {
    uint32_t xa[2] = {C::XABS_LIMBS[0], C::XABS_LIMBS[1]};
    jac_mul_bits(j, q, xa, 64);
    Aff<Fp2<C>> e = q;
    endo_apply<C>(e, 1);  // psi(Q)
    e.y = neg(e.y);       // x < 0: psi(Q) = -[|x|] Q
    return jac_eq_aff(j, e);
  }
}

// Encoded sizes: compressed = NC Fq, uncompressed = 2 NC Fq
template <class C, class F> GS_HD_NOINLINE void wire_encode_point(uint8_t* out, const Aff<F>& p, bool compressed) {
  constexpr int NC = NCoord<F>::V, B = C::N * 4;
  const int total = (compressed ? NC : 2 * NC) * B;
  bool inf = aff_is_inf(p);
  uint32_t xw[2][C::N], yw[2][C::N];
  coord_words<C>(xw, p.x);
  coord_words<C>(yw, p.y);
  bool largest = !inf && y_is_largest<C, NC>(yw);
  if (inf) {
    for (int i = 0; i < total; i++) out[i] = 0;
  } else if (!C::IS_BN) {  // c1 before c0, big-endian
    for (int k = 0; k < NC; k++) fq_put_bytes<C>(out + k * B, xw[NC - 1 - k]);
    if (!compressed)
      for (int k = 0; k < NC; k++) fq_put_bytes<C>(out + (NC + k) * B, yw[NC - 1 - k]);
  } else {
    for (int k = 0; k < NC; k++) fq_put_bytes<C>(out + k * B, xw[k]);
    if (!compressed)
      for (int k = 0; k < NC; k++) fq_put_bytes<C>(out + (NC + k) * B, yw[k]);
  }
  if (!C::IS_BN) {
    uint8_t fl = (compressed ? 0x80 : 0) | (inf ? 0x40 : 0) | ((compressed && largest) ? 0x20 : 0);
    out[0] |= fl;
  } else {
    uint8_t fl = (inf ? 0x40 : 0) | (largest ? 0x80 : 0);
    out[total - 1] |= fl;
  }
}

// returns false on any malformed / invalid input; p is then the identity
template <class C, class F>
GS_HD_NOINLINE bool wire_decode_point(Aff<F>& p, const uint8_t* in, bool compressed, bool validate) {
  constexpr int NC = NCoord<F>::V, B = C::N * 4;
  const int total = (compressed ? NC : 2 * NC) * B;
  p.x = zero_of<F>();
  p.y = zero_of<F>();
  uint8_t fb = C::IS_BN ? in[total - 1] : in[0];
  bool inf, largest;
  uint8_t mask;
  if (!C::IS_BN) {
    mask = 0xE0;
    if (((fb & 0x80) != 0) != compressed) return false;
    inf = fb & 0x40;
    largest = fb & 0x20;
    if (!compressed && largest) return false;
    if (inf && largest) return false;  // ark-bls12-381 EncodingFlags: the sort flag never accompanies infinity
  } else {
    mask = 0xC0;
    inf = fb & 0x40;
    largest = fb & 0x80;
    if (inf && largest) return false;
  }
  const int flag_pos = C::IS_BN ? total - 1 : 0;
  uint32_t xw[2][C::N], yw[2][C::N];
  for (int k = 0; k < NC; k++) {
    int slot = C::IS_BN ? k : NC - 1 - k;
    fq_get_bytes<C>(xw[slot], in + k * B, mask, flag_pos - k * B);
    if (!compressed) fq_get_bytes<C>(yw[slot], in + (NC + k) * B, mask, flag_pos - (NC + k) * B);
  }
  bool canon = true, zero = true;
  for (int k = 0; k < NC; k++) {
    canon = canon && words_lt_p<C>(xw[k]);
    zero = zero && words_zero<C::N>(xw[k]);
    if (!compressed) {
      canon = canon && words_lt_p<C>(yw[k]);
      zero = zero && words_zero<C::N>(yw[k]);
    }
  }
  if (!canon) return false;
  // The identity is all-zero coordinates plus the flag.  Deliberately STRICTER than ark-bls12-381, which returns the
  // identity as soon as it sees the infinity flag and ignores the payload [ark-mem]: only the one encoding arkworks'
  // own serialiser produces is accepted, so no second byte string decodes to the same element (include/gs_amd.h).
  if (inf) return zero;
  F x, y;
  coord_from_words<C>(x, xw);
  F rhs = curve_rhs<C>(x);
  if (compressed) {
    bool sq;
    y = f_sqrt<C>(rhs, sq);
    if (!sq) return false;
    uint32_t got[2][C::N];
    coord_words<C>(got, y);
    if (y_is_largest<C, NC>(got) != largest) y = neg(y);
  } else {
    coord_from_words<C>(y, yw);
    if (!eq(sqr(y), rhs)) return false;
  }
  p.x = x;
  p.y = norm(y);
  if (validate && !in_prime_subgroup<C>(p)) {
    p.x = zero_of<F>();
    p.y = zero_of<F>();
    return false;
  }
  return true;
}

// f^r == 1 for an arbitrary f (not assumed cyclotomic): PairingOutput's validity check
template <class C> GS_HD_NOINLINE bool f12_in_torsion(const Fp12<C>& f) {
  Fp12<C> acc;
  f12_one(acc);
  bool started = false;
  for (int i = FrM<C>::N * 32 - 1; i >= 0; i--) {
    if (started) f12_sqr(acc, acc);
    if ((C::R_WORDS[i >> 5] >> (i & 31)) & 1) {
      if (started) f12_mul(acc, acc, f); else acc = f;
      started = true;
    }
  }
  return f12_is_one(acc);
}

}  // namespace gs
