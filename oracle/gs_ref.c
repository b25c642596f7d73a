/* gs_ref.c -- CPU restatement (plain C, 64-bit limbs) of the reference's
 * Groth-Sahai path, following the REFERENCE'S OWN EVALUATION ORDER.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/ and the `cpu_baseline` leg of
 * bench.py ("kind": "port").  Nothing under groth_sahai_rs_amd/ links or calls it.
 *
 * What it follows (file:line relative to /root/reference):
 *   Com1/Com2 Add/Neg/scalar_mul with per-operation affine normalisation
 *                                   src/data_structures.rs:181-251, 336-342, 381-387
 *   iota maps, W1/W2                src/data_structures.rs:310-334, 355-379
 *   ComT::pairing / pairing_sum (4 multi-pairings = 4 final exponentiations each)
 *                                   src/data_structures.rs:484-502, 509-540
 *   Matrix<Com>::left_mul, Matrix<Fr> products (naive, term by term)
 *                                   src/data_structures.rs:696-742, 824-912
 *   batch_commit_*                  src/prover/commit.rs:78-100,125-156,178-200,225-256
 *   Provable::prove x4              src/prover/prove.rs:92-171,195-274,298-379,409-488
 *   Verifiable::verify x4           src/verifier.rs:23-157   (Gamma * d on the G2 side, five pairing_sums)
 *
 * arkworks (ark-ff / ark-ec ^0.5, NOT in /root/reference, no Cargo.lock) is
 * restated from its published algorithms: Montgomery CIOS, Karatsuba towers,
 * Jacobian arithmetic, plain double-and-add `mul_bigint`, the projective Miller
 * loop with `ell` line evaluation, and the eprint 2020/875 final exponentiation
 * (BLS12) / Fuentes-Castaneda hard part (BN).  PARITY STATUS: pinned only by the
 * big-integer oracle's fixtures (tests/golden), i.e. "parity unpinned" at the
 * byte level w.r.t. arkworks -- see oracle/gs_oracle.py header.
 *
 * Build: oracle/Makefile compiles this file once per curve
 *   gcc -O2 -DCURVE_BLS12_381 ... -> oracle/libgs_ref_bls12_381.so
 */
#include <pthread.h>
#include <unistd.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#if defined(CURVE_BN254)
#include "gs_ref_params_bn254.h"
#else
#include "gs_ref_params_bls12_381.h"
#endif

/* ------------------------------------------------------------------ Fp --- */
typedef struct { u64 l[NL]; } fp;
typedef struct { u64 l[4]; } fr;

/* CIOS Montgomery product; `n` is a compile-time constant at every call site
 * (always_inline + full unrolling), u128 accumulators -> mulx/adcx-class code. */
static inline __attribute__((always_inline)) void mont_mul_n(u64* r, const u64* a, const u64* b, const u64* mod,
                                                             u64 inv, const int n) {
  u64 t[8 + 2];
#pragma GCC unroll 10
  for (int i = 0; i < n + 2; i++) t[i] = 0;
#pragma GCC unroll 8
  for (int i = 0; i < n; i++) {
    u128 c = 0;
#pragma GCC unroll 8
    for (int j = 0; j < n; j++) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (u64)c;
      c >>= 64;
    }
    c += t[n];
    t[n] = (u64)c;
    t[n + 1] = (u64)(c >> 64);
    u64 m = t[0] * inv;
    c = (u128)m * mod[0] + t[0];
    c >>= 64;
#pragma GCC unroll 8
    for (int j = 1; j < n; j++) {
      c += (u128)m * mod[j] + t[j];
      t[j - 1] = (u64)c;
      c >>= 64;
    }
    c += t[n];
    t[n - 1] = (u64)c;
    t[n] = t[n + 1] + (u64)(c >> 64);
  }
  u64 d[8];
  u64 br = 0;
#pragma GCC unroll 8
  for (int j = 0; j < n; j++) {
    u128 s = (u128)t[j] - mod[j] - br;
    d[j] = (u64)s;
    br = (u64)(s >> 127);
  }
  int keep = (t[n] == 0) && br;
#pragma GCC unroll 8
  for (int j = 0; j < n; j++) r[j] = keep ? t[j] : d[j];
}
static inline __attribute__((always_inline)) void add_n(u64* r, const u64* a, const u64* b, const u64* mod, int n) {
  u64 s[8], d[8];
  u64 c = 0, br = 0;
  for (int j = 0; j < n; j++) {
    u128 x = (u128)a[j] + b[j] + c;
    s[j] = (u64)x;
    c = (u64)(x >> 64);
  }
  for (int j = 0; j < n; j++) {
    u128 x = (u128)s[j] - mod[j] - br;
    d[j] = (u64)x;
    br = (u64)(x >> 127);
  }
  int keep = br && !c;
  for (int j = 0; j < n; j++) r[j] = keep ? s[j] : d[j];
}
static inline __attribute__((always_inline)) void sub_n(u64* r, const u64* a, const u64* b, const u64* mod, int n) {
  u64 d[8];
  u64 br = 0, c = 0;
  for (int j = 0; j < n; j++) {
    u128 x = (u128)a[j] - b[j] - br;
    d[j] = (u64)x;
    br = (u64)(x >> 127);
  }
  for (int j = 0; j < n; j++) {
    u128 x = (u128)d[j] + (br ? mod[j] : 0) + c;
    r[j] = (u64)x;
    c = (u64)(x >> 64);
  }
}
static int is_zero_n(const u64* a, int n) {
  u64 o = 0;
  for (int j = 0; j < n; j++) o |= a[j];
  return o == 0;
}

static __thread u64 g_fpmul_count; /* instrumented Fp-multiplication counter */

static void fp_mul(fp* r, const fp* a, const fp* b) {
  g_fpmul_count++;
  mont_mul_n(r->l, a->l, b->l, P_MOD, P_INV, NL);
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
static void fp_add(fp* r, const fp* a, const fp* b) { add_n(r->l, a->l, b->l, P_MOD, NL); }
static void fp_sub(fp* r, const fp* a, const fp* b) { sub_n(r->l, a->l, b->l, P_MOD, NL); }
static void fp_dbl(fp* r, const fp* a) { fp_add(r, a, a); }
static int fp_is_zero(const fp* a) { return is_zero_n(a->l, NL); }
static int fp_eq(const fp* a, const fp* b) { return memcmp(a, b, sizeof(fp)) == 0; }
static void fp_zero(fp* r) { memset(r, 0, sizeof *r); }
static void fp_one(fp* r) { memcpy(r->l, P_ONE, sizeof r->l); }
static void fp_neg(fp* r, const fp* a) {
  if (fp_is_zero(a)) {
    *r = *a;
    return;
  }
  fp z;
  fp_zero(&z);
  fp_sub(r, &z, a);
}
static void fp_inv(fp* r, const fp* a) { /* a^(p-2), plain square-and-multiply */
  u64 e[NL];
  memcpy(e, P_MOD, sizeof e);
  e[0] -= 2;
  fp acc, base = *a;
  fp_one(&acc);
  int started = 0;
  for (int i = NL * 64 - 1; i >= 0; i--) {
    if (started) fp_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) {
      if (started) fp_mul(&acc, &acc, &base);
      else acc = base;
      started = 1;
    }
  }
  *r = acc;
}

static void fr_mul(fr* r, const fr* a, const fr* b) { mont_mul_n(r->l, a->l, b->l, Q_MOD, Q_INV, 4); }
static void fr_add(fr* r, const fr* a, const fr* b) { add_n(r->l, a->l, b->l, Q_MOD, 4); }
static void fr_sub(fr* r, const fr* a, const fr* b) { sub_n(r->l, a->l, b->l, Q_MOD, 4); }
static void fr_neg(fr* r, const fr* a) {
  fr z;
  memset(&z, 0, sizeof z);
  if (is_zero_n(a->l, 4)) *r = *a;
  else fr_sub(r, &z, a);
}
static void fr_from_mont(u64* out, const fr* a) {
  fr one;
  memset(&one, 0, sizeof one);
  one.l[0] = 1;
  fr t;
  fr_mul(&t, a, &one);
  memcpy(out, t.l, 32);
}
static void fr_to_mont(fr* r, const u64* canon) {
  fr t, r2;
  memcpy(t.l, canon, 32);
  memcpy(r2.l, Q_R2, 32);
  fr_mul(r, &t, &r2);
}

/* ------------------------------------------------------------------ Fp2 -- */
typedef struct { fp c0, c1; } fp2;
static void f2_add(fp2* r, const fp2* a, const fp2* b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static void f2_sub(fp2* r, const fp2* a, const fp2* b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static void f2_neg(fp2* r, const fp2* a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static void f2_dbl(fp2* r, const fp2* a) { f2_add(r, a, a); }
static void f2_conj(fp2* r, const fp2* a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
static int f2_is_zero(const fp2* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int f2_eq(const fp2* a, const fp2* b) { return memcmp(a, b, sizeof(fp2)) == 0; }
static void f2_zero(fp2* r) { memset(r, 0, sizeof *r); }
static void f2_one(fp2* r) { fp_one(&r->c0); fp_zero(&r->c1); }
static void f2_mul(fp2* r, const fp2* a, const fp2* b) {
  fp v0, v1, s, t;
  fp_mul(&v0, &a->c0, &b->c0);
  fp_mul(&v1, &a->c1, &b->c1);
  fp_add(&s, &a->c0, &a->c1);
  fp_add(&t, &b->c0, &b->c1);
  fp_mul(&s, &s, &t);
  fp_sub(&r->c0, &v0, &v1);
  fp_sub(&s, &s, &v0);
  fp_sub(&r->c1, &s, &v1);
}
static void f2_sqr(fp2* r, const fp2* a) {
  fp s, d, t;
  fp_add(&s, &a->c0, &a->c1);
  fp_sub(&d, &a->c0, &a->c1);
  fp_mul(&t, &a->c0, &a->c1);
  fp_mul(&r->c0, &s, &d);
  fp_dbl(&r->c1, &t);
}
static void f2_mul_fp(fp2* r, const fp2* a, const fp* k) { fp_mul(&r->c0, &a->c0, k); fp_mul(&r->c1, &a->c1, k); }
static void f2_mul_xi(fp2* r, const fp2* a) {
  fp t0 = a->c0, t1 = a->c1, x0, x1;
#if XI_A == 1
  fp_sub(&x0, &t0, &t1);
  fp_add(&x1, &t0, &t1);
#else
  fp n0, n1;
  fp_dbl(&n0, &t0); fp_dbl(&n0, &n0); fp_dbl(&n0, &n0); fp_add(&n0, &n0, &t0); /* 9 a0 */
  fp_dbl(&n1, &t1); fp_dbl(&n1, &n1); fp_dbl(&n1, &n1); fp_add(&n1, &n1, &t1); /* 9 a1 */
  fp_sub(&x0, &n0, &t1);
  fp_add(&x1, &n1, &t0);
#endif
  r->c0 = x0;
  r->c1 = x1;
}
static void f2_inv(fp2* r, const fp2* a) {
  fp n, t;
  fp_sqr(&n, &a->c0);
  fp_sqr(&t, &a->c1);
  fp_add(&n, &n, &t);
  fp_inv(&n, &n);
  fp_mul(&r->c0, &a->c0, &n);
  fp_mul(&t, &a->c1, &n);
  fp_neg(&r->c1, &t);
}

/* ------------------------------------------------------------------ Fp6 -- */
typedef struct { fp2 c0, c1, c2; } fp6;
static void f6_add(fp6* r, const fp6* a, const fp6* b) { f2_add(&r->c0, &a->c0, &b->c0); f2_add(&r->c1, &a->c1, &b->c1); f2_add(&r->c2, &a->c2, &b->c2); }
static void f6_sub(fp6* r, const fp6* a, const fp6* b) { f2_sub(&r->c0, &a->c0, &b->c0); f2_sub(&r->c1, &a->c1, &b->c1); f2_sub(&r->c2, &a->c2, &b->c2); }
static void f6_neg(fp6* r, const fp6* a) { f2_neg(&r->c0, &a->c0); f2_neg(&r->c1, &a->c1); f2_neg(&r->c2, &a->c2); }
static void f6_mul_v(fp6* r, const fp6* a) {
  fp2 t;
  f2_mul_xi(&t, &a->c2);
  r->c2 = a->c1;
  r->c1 = a->c0;
  r->c0 = t;
}
static void f6_mul(fp6* r, const fp6* a, const fp6* b) {
  fp2 v0, v1, v2, t0, t1, t2, s, u;
  f2_mul(&v0, &a->c0, &b->c0);
  f2_mul(&v1, &a->c1, &b->c1);
  f2_mul(&v2, &a->c2, &b->c2);
  f2_add(&s, &a->c1, &a->c2); f2_add(&u, &b->c1, &b->c2); f2_mul(&t0, &s, &u); f2_sub(&t0, &t0, &v1); f2_sub(&t0, &t0, &v2);
  f2_add(&s, &a->c0, &a->c1); f2_add(&u, &b->c0, &b->c1); f2_mul(&t1, &s, &u); f2_sub(&t1, &t1, &v0); f2_sub(&t1, &t1, &v1);
  f2_add(&s, &a->c0, &a->c2); f2_add(&u, &b->c0, &b->c2); f2_mul(&t2, &s, &u); f2_sub(&t2, &t2, &v0); f2_sub(&t2, &t2, &v2);
  f2_mul_xi(&t0, &t0);
  f2_add(&r->c0, &v0, &t0);
  f2_mul_xi(&s, &v2);
  f2_add(&r->c1, &t1, &s);
  f2_add(&r->c2, &t2, &v1);
}
static void f6_inv(fp6* r, const fp6* a) {
  fp2 t0, t1, t2, s, n;
  f2_sqr(&t0, &a->c0); f2_mul(&s, &a->c1, &a->c2); f2_mul_xi(&s, &s); f2_sub(&t0, &t0, &s);
  f2_sqr(&t1, &a->c2); f2_mul_xi(&t1, &t1); f2_mul(&s, &a->c0, &a->c1); f2_sub(&t1, &t1, &s);
  f2_sqr(&t2, &a->c1); f2_mul(&s, &a->c0, &a->c2); f2_sub(&t2, &t2, &s);
  f2_mul(&n, &a->c2, &t1); f2_mul(&s, &a->c1, &t2); f2_add(&n, &n, &s); f2_mul_xi(&n, &n);
  f2_mul(&s, &a->c0, &t0); f2_add(&n, &n, &s);
  f2_inv(&n, &n);
  f2_mul(&r->c0, &t0, &n);
  f2_mul(&r->c1, &t1, &n);
  f2_mul(&r->c2, &t2, &n);
}
/* sparse products used by the line evaluation (ark-ff fp6_3over2 mul_by_01 / mul_by_1) */
static void f6_mul_by_01(fp6* r, const fp6* a, const fp2* b0, const fp2* b1) {
  fp6 b;
  b.c0 = *b0; b.c1 = *b1; f2_zero(&b.c2);
  f6_mul(r, a, &b);
}
static void f6_mul_by_1(fp6* r, const fp6* a, const fp2* b1) {
  fp6 b;
  f2_zero(&b.c0); b.c1 = *b1; f2_zero(&b.c2);
  f6_mul(r, a, &b);
}

/* ----------------------------------------------------------------- Fp12 -- */
typedef struct { fp6 c0, c1; } fp12;
static void f12_one(fp12* r) { memset(r, 0, sizeof *r); fp_one(&r->c0.c0.c0); }
static int f12_eq(const fp12* a, const fp12* b) { return memcmp(a, b, sizeof(fp12)) == 0; }
static void f12_mul(fp12* r, const fp12* a, const fp12* b) {
  fp6 t0, t1, sa, sb, m;
  f6_mul(&t0, &a->c0, &b->c0);
  f6_mul(&t1, &a->c1, &b->c1);
  f6_add(&sa, &a->c0, &a->c1);
  f6_add(&sb, &b->c0, &b->c1);
  f6_mul(&m, &sa, &sb);
  f6_sub(&m, &m, &t0);
  f6_sub(&r->c1, &m, &t1);
  f6_mul_v(&t1, &t1);
  f6_add(&r->c0, &t0, &t1);
}
static void f12_sqr(fp12* r, const fp12* a) { f12_mul(r, a, a); }
static void f12_conj(fp12* r, const fp12* a) { r->c0 = a->c0; f6_neg(&r->c1, &a->c1); }
static void f12_inv(fp12* r, const fp12* a) {
  fp6 t0, t1;
  f6_mul(&t0, &a->c0, &a->c0);
  f6_mul(&t1, &a->c1, &a->c1);
  f6_mul_v(&t1, &t1);
  f6_sub(&t0, &t0, &t1);
  f6_inv(&t1, &t0);
  f6_mul(&r->c0, &a->c0, &t1);
  f6_mul(&t0, &a->c1, &t1);
  f6_neg(&r->c1, &t0);
}
static void f12_frob(fp12* r, const fp12* a, int j) {
  const u64(*T)[2][NL] = j == 1 ? FROB1 : j == 2 ? FROB2 : FROB3;
  const fp2* src[6] = {&a->c0.c0, &a->c1.c0, &a->c0.c1, &a->c1.c1, &a->c0.c2, &a->c1.c2}; /* w^0..w^5 */
  fp2* dst[6] = {&r->c0.c0, &r->c1.c0, &r->c0.c1, &r->c1.c1, &r->c0.c2, &r->c1.c2};
  fp12 out;
  fp2* od[6] = {&out.c0.c0, &out.c1.c0, &out.c0.c1, &out.c1.c1, &out.c0.c2, &out.c1.c2};
  for (int k = 0; k < 6; k++) {
    fp2 x = *src[k], g;
    if (j & 1) f2_conj(&x, &x);
    memcpy(&g, T[k], sizeof g);
    f2_mul(od[k], &x, &g);
  }
  (void)dst;
  *r = out;
}
static void f12_mul_by_014(fp12* f, const fp2* c0, const fp2* c1, const fp2* c4) {
  fp6 aa, bb, s, t;
  fp2 o;
  f6_mul_by_01(&aa, &f->c0, c0, c1);
  f6_mul_by_1(&bb, &f->c1, c4);
  f2_add(&o, c1, c4);
  f6_add(&s, &f->c1, &f->c0);
  f6_mul_by_01(&t, &s, c0, &o);
  f6_sub(&t, &t, &aa);
  f6_sub(&f->c1, &t, &bb);
  f6_mul_v(&bb, &bb);
  f6_add(&f->c0, &bb, &aa);
}
static void f12_mul_by_034(fp12* f, const fp2* c0, const fp2* c3, const fp2* c4) {
  fp6 a, b, s, t;
  fp2 o;
  f2_mul(&a.c0, &f->c0.c0, c0); f2_mul(&a.c1, &f->c0.c1, c0); f2_mul(&a.c2, &f->c0.c2, c0);
  f6_mul_by_01(&b, &f->c1, c3, c4);
  f2_add(&o, c0, c3);
  f6_add(&s, &f->c0, &f->c1);
  f6_mul_by_01(&t, &s, &o, c4);
  f6_sub(&t, &t, &a);
  f6_sub(&f->c1, &t, &b);
  f6_mul_v(&b, &b);
  f6_add(&f->c0, &a, &b);
}
static void f12_exp_by_x(fp12* r, const fp12* f) { /* f^x (signed x), plain squarings */
  fp12 acc = *f;
  int top = 63;
  while (!((X_ABS >> top) & 1)) top--;
  for (int i = top - 1; i >= 0; i--) {
    f12_sqr(&acc, &acc);
    if ((X_ABS >> i) & 1) f12_mul(&acc, &acc, f);
  }
#if X_NEG
  f12_conj(&acc, &acc);
#endif
  *r = acc;
}

/* ------------------------------------------------------------ G1 / G2 --- */
typedef struct { fp x, y; } g1a;     /* affine; identity = all zero */
typedef struct { fp2 x, y; } g2a;
typedef struct { fp x, y, z; } g1j;  /* Jacobian; identity z = 0 */
typedef struct { fp2 x, y, z; } g2j;

#define DEFINE_GROUP(G, F, FADD, FSUB, FMUL, FSQR, FDBL, FNEG, FISZ, FEQ, FINV, FONE, FZERO)                 \
  static int G##a_is_inf(const G##a* p) { return FISZ(&p->x) && FISZ(&p->y); }                              \
  static void G##j_inf(G##j* r) { FONE(&r->x); FONE(&r->y); FZERO(&r->z); }                                 \
  static void G##_to_jac(G##j* r, const G##a* p) {                                                          \
    if (G##a_is_inf(p)) { G##j_inf(r); return; }                                                            \
    r->x = p->x; r->y = p->y; FONE(&r->z);                                                                  \
  }                                                                                                         \
  static void G##j_dbl(G##j* r, const G##j* p) {                                                            \
    if (FISZ(&p->z)) { *r = *p; return; }                                                                   \
    F a, b, c, d, e, f, t, x3, y3, z3;                                                                      \
    FSQR(&a, &p->x); FSQR(&b, &p->y); FSQR(&c, &b);                                                         \
    FADD(&t, &p->x, &b); FSQR(&t, &t); FSUB(&t, &t, &a); FSUB(&t, &t, &c); FDBL(&d, &t);                    \
    FDBL(&e, &a); FADD(&e, &e, &a); FSQR(&f, &e);                                                           \
    FMUL(&z3, &p->y, &p->z); FDBL(&z3, &z3);                                                                \
    FDBL(&t, &d); FSUB(&x3, &f, &t);                                                                        \
    FSUB(&t, &d, &x3); FMUL(&y3, &e, &t); FDBL(&c, &c); FDBL(&c, &c); FDBL(&c, &c); FSUB(&y3, &y3, &c);     \
    r->x = x3; r->y = y3; r->z = z3;                                                                        \
  }                                                                                                         \
  static void G##j_add(G##j* r, const G##j* p, const G##j* q) {                                             \
    if (FISZ(&p->z)) { *r = *q; return; }                                                                   \
    if (FISZ(&q->z)) { *r = *p; return; }                                                                   \
    F z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;                                            \
    FSQR(&z1z1, &p->z); FSQR(&z2z2, &q->z);                                                                 \
    FMUL(&u1, &p->x, &z2z2); FMUL(&u2, &q->x, &z1z1);                                                       \
    FMUL(&s1, &p->y, &q->z); FMUL(&s1, &s1, &z2z2);                                                         \
    FMUL(&s2, &q->y, &p->z); FMUL(&s2, &s2, &z1z1);                                                         \
    if (FEQ(&u1, &u2)) {                                                                                    \
      if (FEQ(&s1, &s2)) { G##j_dbl(r, p); return; }                                                        \
      G##j_inf(r); return;                                                                                  \
    }                                                                                                       \
    FSUB(&h, &u2, &u1); FDBL(&i, &h); FSQR(&i, &i); FMUL(&j, &h, &i);                                       \
    FSUB(&rr, &s2, &s1); FDBL(&rr, &rr); FMUL(&v, &u1, &i);                                                 \
    FSQR(&x3, &rr); FSUB(&x3, &x3, &j); FDBL(&t, &v); FSUB(&x3, &x3, &t);                                   \
    FSUB(&t, &v, &x3); FMUL(&y3, &rr, &t); FMUL(&t, &s1, &j); FDBL(&t, &t); FSUB(&y3, &y3, &t);             \
    FADD(&z3, &p->z, &q->z); FSQR(&z3, &z3); FSUB(&z3, &z3, &z1z1); FSUB(&z3, &z3, &z2z2); FMUL(&z3, &z3, &h); \
    r->x = x3; r->y = y3; r->z = z3;                                                                        \
  }                                                                                                         \
  /* into_affine: one inversion per call, as the reference does per operation */                            \
  static void G##_to_aff(G##a* r, const G##j* p) {                                                          \
    if (FISZ(&p->z)) { memset(r, 0, sizeof *r); return; }                                                   \
    F zi, zi2, zi3;                                                                                         \
    FINV(&zi, &p->z); FSQR(&zi2, &zi); FMUL(&zi3, &zi2, &zi);                                               \
    FMUL(&r->x, &p->x, &zi2); FMUL(&r->y, &p->y, &zi3);                                                     \
  }                                                                                                         \
  /* `Projective *= scalar`: plain MSB-first double-and-add over the canonical bits [ark-mem] */            \
  static void G##_mul(G##a* r, const G##a* p, const fr* k_mont) {                                           \
    u64 k[4];                                                                                               \
    fr_from_mont(k, k_mont);                                                                                \
    G##j acc, base;                                                                                         \
    G##j_inf(&acc);                                                                                         \
    G##_to_jac(&base, p);                                                                                   \
    int started = 0;                                                                                        \
    for (int i = 255; i >= 0; i--) {                                                                        \
      if (started) G##j_dbl(&acc, &acc);                                                                    \
      if ((k[i >> 6] >> (i & 63)) & 1) { G##j_add(&acc, &acc, &base); started = 1; }                        \
    }                                                                                                       \
    G##_to_aff(r, &acc);                                                                                    \
  }                                                                                                         \
  /* affine + affine -> projective add -> .into() affine (data_structures.rs:187-188) */                   \
  static void G##_add(G##a* r, const G##a* p, const G##a* q) {                                              \
    G##j a, b;                                                                                              \
    G##_to_jac(&a, p); G##_to_jac(&b, q);                                                                   \
    G##j_add(&a, &a, &b);                                                                                   \
    G##_to_aff(r, &a);                                                                                      \
  }                                                                                                         \
  static void G##_neg(G##a* r, const G##a* p) { r->x = p->x; FNEG(&r->y, &p->y); }

DEFINE_GROUP(g1, fp, fp_add, fp_sub, fp_mul, fp_sqr, fp_dbl, fp_neg, fp_is_zero, fp_eq, fp_inv, fp_one, fp_zero)
DEFINE_GROUP(g2, fp2, f2_add, f2_sub, f2_mul, f2_sqr, f2_dbl, f2_neg, f2_is_zero, f2_eq, f2_inv, f2_one, f2_zero)

/* ------------------------------------------------------------- pairing --- */
typedef struct { fp2 x, y, z; } g2h; /* homogeneous projective */
typedef struct { fp2 c0, c1, c2; } ellc;

static void ell_double(g2h* t, ellc* o) { /* ark-ec bls12/g2.rs double_in_place [ark-mem] */
  fp2 a, b, c, e, f, g, h, i, j, e2, tmp, b3;
  fp two_inv;
  memcpy(two_inv.l, TWO_INV, sizeof two_inv.l);
  memcpy(&b3, COEFF_B2, sizeof b3);
  f2_mul(&a, &t->x, &t->y); f2_mul_fp(&a, &a, &two_inv);
  f2_sqr(&b, &t->y);
  f2_sqr(&c, &t->z);
  f2_dbl(&tmp, &c); f2_add(&tmp, &tmp, &c); f2_mul(&e, &b3, &tmp);
  f2_dbl(&f, &e); f2_add(&f, &f, &e);
  f2_add(&g, &b, &f); f2_mul_fp(&g, &g, &two_inv);
  f2_add(&h, &t->y, &t->z); f2_sqr(&h, &h); f2_add(&tmp, &b, &c); f2_sub(&h, &h, &tmp);
  f2_sub(&i, &e, &b);
  f2_sqr(&j, &t->x);
  f2_sqr(&e2, &e);
  f2_sub(&tmp, &b, &f); f2_mul(&t->x, &a, &tmp);
  f2_sqr(&tmp, &g); { fp2 e3; f2_dbl(&e3, &e2); f2_add(&e3, &e3, &e2); f2_sub(&t->y, &tmp, &e3); }
  f2_mul(&t->z, &b, &h);
  fp2 j3, nh;
  f2_dbl(&j3, &j); f2_add(&j3, &j3, &j);
  f2_neg(&nh, &h);
#if TWIST_M
  o->c0 = i; o->c1 = j3; o->c2 = nh;
#else
  o->c0 = nh; o->c1 = j3; o->c2 = i;
#endif
}
static void ell_add(g2h* t, const g2a* q, ellc* o) { /* add_in_place [ark-mem] */
  fp2 theta, lambda, c, d, e, f, g, h, j, tmp;
  f2_mul(&tmp, &q->y, &t->z); f2_sub(&theta, &t->y, &tmp);
  f2_mul(&tmp, &q->x, &t->z); f2_sub(&lambda, &t->x, &tmp);
  f2_sqr(&c, &theta); f2_sqr(&d, &lambda); f2_mul(&e, &lambda, &d);
  f2_mul(&f, &t->z, &c); f2_mul(&g, &t->x, &d);
  f2_add(&h, &e, &f); f2_dbl(&tmp, &g); f2_sub(&h, &h, &tmp);
  f2_mul(&t->x, &lambda, &h);
  f2_sub(&tmp, &g, &h); f2_mul(&tmp, &theta, &tmp); { fp2 ey; f2_mul(&ey, &e, &t->y); f2_sub(&t->y, &tmp, &ey); }
  f2_mul(&t->z, &t->z, &e);
  { fp2 a, b; f2_mul(&a, &theta, &q->x); f2_mul(&b, &lambda, &q->y); f2_sub(&j, &a, &b); }
  fp2 nt;
  f2_neg(&nt, &theta);
#if TWIST_M
  o->c0 = j; o->c1 = nt; o->c2 = lambda;
#else
  o->c0 = lambda; o->c1 = nt; o->c2 = j;
#endif
}
static void ell_eval(fp12* f, const ellc* co, const g1a* p) { /* ark-ec `ell` */
  fp2 c0 = co->c0, c1 = co->c1, c2 = co->c2;
#if TWIST_M
  f2_mul_fp(&c2, &c2, &p->y);
  f2_mul_fp(&c1, &c1, &p->x);
  f12_mul_by_014(f, &c0, &c1, &c2);
#else
  f2_mul_fp(&c0, &c0, &p->y);
  f2_mul_fp(&c1, &c1, &p->x);
  f12_mul_by_034(f, &c0, &c1, &c2);
#endif
}
/* multi_miller_loop: pairs with an identity argument are skipped */
static void multi_miller(fp12* f, const g1a* ps, const g2a* qs, int n) {
  f12_one(f);
  int nl = 0;
  g2h* ts = (g2h*)malloc(sizeof(g2h) * (n ? n : 1));
  int* idx = (int*)malloc(sizeof(int) * (n ? n : 1));
  for (int k = 0; k < n; k++) {
    if (g1a_is_inf(&ps[k]) || g2a_is_inf(&qs[k])) continue;
    ts[nl].x = qs[k].x; ts[nl].y = qs[k].y; f2_one(&ts[nl].z);
    idx[nl++] = k;
  }
  ellc co;
  for (int i = LOOP_LEN - 2; i >= 0; i--) {
    f12_sqr(f, f);
    for (int k = 0; k < nl; k++) { ell_double(&ts[k], &co); ell_eval(f, &co, &ps[idx[k]]); }
    int d = LOOP[i];
    if (d) for (int k = 0; k < nl; k++) {
      g2a q = qs[idx[k]];
      if (d < 0) f2_neg(&q.y, &q.y);
      ell_add(&ts[k], &q, &co); ell_eval(f, &co, &ps[idx[k]]);
    }
  }
#if IS_BN
  for (int k = 0; k < nl; k++) {
    g2a q1, q2;
    fp2 g;
    const g2a* q = &qs[idx[k]];
    f2_conj(&q1.x, &q->x); memcpy(&g, FROB1[2], sizeof g); f2_mul(&q1.x, &q1.x, &g);
    f2_conj(&q1.y, &q->y); memcpy(&g, FROB1[3], sizeof g); f2_mul(&q1.y, &q1.y, &g);
    memcpy(&g, FROB2[2], sizeof g); f2_mul(&q2.x, &q->x, &g);
    memcpy(&g, FROB2[3], sizeof g); f2_mul(&q2.y, &q->y, &g); f2_neg(&q2.y, &q2.y);
    ell_add(&ts[k], &q1, &co); ell_eval(f, &co, &ps[idx[k]]);
    ell_add(&ts[k], &q2, &co); ell_eval(f, &co, &ps[idx[k]]);
  }
#endif
#if X_NEG && !IS_BN
  f12_conj(f, f);
#endif
  free(ts);
  free(idx);
}
static void final_exp(fp12* out, const fp12* fin) {
  fp12 f = *fin, f1, f2, r, y0, y1, y2;
  f12_conj(&f1, &f);
  f12_inv(&f2, &f);
  f12_mul(&r, &f1, &f2);
  f2 = r;
  f12_frob(&r, &r, 2);
  f12_mul(&r, &r, &f2);
#if !IS_BN
  f12_sqr(&y0, &r);
  f12_exp_by_x(&y1, &r);
  f12_conj(&y2, &r);
  f12_mul(&y1, &y1, &y2);
  f12_exp_by_x(&y2, &y1);
  f12_conj(&y1, &y1);
  f12_mul(&y1, &y1, &y2);
  f12_exp_by_x(&y2, &y1);
  f12_frob(&y1, &y1, 1);
  f12_mul(&y1, &y1, &y2);
  f12_mul(&r, &r, &y0);
  f12_exp_by_x(&y0, &y1);
  f12_exp_by_x(&y2, &y0);
  y0 = y1;
  f12_frob(&y0, &y0, 2);
  f12_conj(&y1, &y1);
  f12_mul(&y1, &y1, &y2);
  f12_mul(&y1, &y1, &y0);
  f12_mul(out, &r, &y1);
#else
  fp12 y3, y4, y5, y6, y7, y8, y9, y10, y11, y12, y13, y14, y15;
  f12_exp_by_x(&y0, &r); f12_conj(&y0, &y0);
  f12_sqr(&y1, &y0);
  f12_sqr(&y2, &y1);
  f12_mul(&y3, &y2, &y1);
  f12_exp_by_x(&y4, &y3); f12_conj(&y4, &y4);
  f12_sqr(&y5, &y4);
  f12_exp_by_x(&y6, &y5); f12_conj(&y6, &y6);
  f12_conj(&y3, &y3);
  f12_conj(&y6, &y6);
  f12_mul(&y7, &y6, &y4);
  f12_mul(&y8, &y7, &y3);
  f12_mul(&y9, &y8, &y1);
  f12_mul(&y10, &y8, &y4);
  f12_mul(&y11, &y10, &r);
  y12 = y9; f12_frob(&y12, &y12, 1);
  f12_mul(&y13, &y12, &y11);
  f12_frob(&y8, &y8, 2);
  f12_mul(&y14, &y8, &y13);
  f12_conj(&r, &r);
  f12_mul(&y15, &r, &y9);
  f12_frob(&y15, &y15, 3);
  f12_mul(out, &y15, &y14);
#endif
}
static void multi_pairing(fp12* out, const g1a* ps, const g2a* qs, int n) {
  fp12 f;
  multi_miller(&f, ps, qs, n);
  final_exp(out, &f);
}

/* ------------------------------------------------- GS commitment group --- */
typedef struct { g1a a, b; } com1;
typedef struct { g2a a, b; } com2;
typedef struct { fp12 c[4]; } comt;
typedef struct { com1 u[2]; com2 v[2]; g1a g1; g2a g2; fp12 gt; } crs_t;

static void com1_add(com1* r, const com1* x, const com1* y) { g1_add(&r->a, &x->a, &y->a); g1_add(&r->b, &x->b, &y->b); }
static void com2_add(com2* r, const com2* x, const com2* y) { g2_add(&r->a, &x->a, &y->a); g2_add(&r->b, &x->b, &y->b); }
static void com1_smul(com1* r, const com1* x, const fr* s) { g1_mul(&r->a, &x->a, s); g1_mul(&r->b, &x->b, s); }
static void com2_smul(com2* r, const com2* x, const fr* s) { g2_mul(&r->a, &x->a, s); g2_mul(&r->b, &x->b, s); }
static void lin1(com1* r, const g1a* x) { memset(&r->a, 0, sizeof r->a); r->b = *x; }
static void lin2(com2* r, const g2a* y) { memset(&r->a, 0, sizeof r->a); r->b = *y; }
static void slin1(com1* r, const fr* x, const crs_t* k) { com1 w, l; lin1(&l, &k->g1); com1_add(&w, &k->u[1], &l); com1_smul(r, &w, x); }
static void slin2(com2* r, const fr* y, const crs_t* k) { com2 w, l; lin2(&l, &k->g2); com2_add(&w, &k->v[1], &l); com2_smul(r, &w, y); }

/* out[i] = sum_k lhs[i*cols+k] * col[k]   (left_mul, sequential branch :730-741) */
static void com1_left_mul(com1* out, const fr* lhs, int rows, int cols, const com1* col) {
  for (int i = 0; i < rows; i++) {
    com1 acc;
    memset(&acc, 0, sizeof acc);
    for (int k = 0; k < cols; k++) { com1 t; com1_smul(&t, &col[k], &lhs[i * cols + k]); com1_add(&acc, &acc, &t); }
    out[i] = acc;
  }
}
static void com2_left_mul_rows(com2* out, const fr* lhs, int r0, int r1, int cols, const com2* col) {
  for (int i = r0; i < r1; i++) {
    com2 acc;
    memset(&acc, 0, sizeof acc);
    for (int k = 0; k < cols; k++) { com2 t; com2_smul(&t, &col[k], &lhs[i * cols + k]); com2_add(&acc, &acc, &t); }
    out[i] = acc;
  }
}
/* Large statements (the reference's 334 x 334 bench shape, benches/bench.rs:451-498): the rows of Gamma * d are
 * independent, exactly as in the reference's Rayon branch (data_structures.rs:707-729) -- same per-term evaluation,
 * rows spread over the host cores so that the test finishes in seconds instead of minutes. */
typedef struct { com2* out; const fr* lhs; int r0, r1, cols; const com2* col; } lm2_job;
static void* lm2_worker(void* a) { lm2_job* j = (lm2_job*)a; com2_left_mul_rows(j->out, j->lhs, j->r0, j->r1, j->cols, j->col); return 0; }
static void com2_left_mul(com2* out, const fr* lhs, int rows, int cols, const com2* col) {
  long work = (long)rows * cols;
  int nt = 1;
  if (work >= 2048) {
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    nt = (int)(nc < 1 ? 1 : nc > 32 ? 32 : nc);
    if (nt > rows) nt = rows;
  }
  if (nt <= 1) { com2_left_mul_rows(out, lhs, 0, rows, cols, col); return; }
  pthread_t th[32];
  lm2_job jobs[32];
  for (int t = 0; t < nt; t++) {
    jobs[t] = (lm2_job){out, lhs, (int)((long)rows * t / nt), (int)((long)rows * (t + 1) / nt), cols, col};
    pthread_create(&th[t], 0, lm2_worker, &jobs[t]);
  }
  for (int t = 0; t < nt; t++) pthread_join(th[t], 0);
}
static void fr_matmul(fr* out, const fr* a, int ar, int ac, const fr* b, int bc) { /* (ar x ac)(ac x bc) */
  for (int i = 0; i < ar; i++) for (int j = 0; j < bc; j++) {
    fr s; memset(&s, 0, sizeof s);
    for (int k = 0; k < ac; k++) { fr t; fr_mul(&t, &a[i * ac + k], &b[k * bc + j]); fr_add(&s, &s, &t); }
    out[i * bc + j] = s;
  }
}
static void fr_transpose(fr* out, const fr* a, int r, int c) { for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) out[j * r + i] = a[i * c + j]; }

static void comt_pairing(comt* r, const com1* x, const com2* y) {
  multi_pairing(&r->c[0], &x->a, &y->a, 1);
  multi_pairing(&r->c[1], &x->a, &y->b, 1);
  multi_pairing(&r->c[2], &x->b, &y->a, 1);
  multi_pairing(&r->c[3], &x->b, &y->b, 1);
}
static void comt_pairing_sum(comt* r, const com1* xs, const com2* ys, int n) {
  g1a* p = (g1a*)malloc(sizeof(g1a) * (n ? n : 1));
  g2a* q = (g2a*)malloc(sizeof(g2a) * (n ? n : 1));
  for (int cell = 0; cell < 4; cell++) {
    for (int k = 0; k < n; k++) { p[k] = (cell >> 1) ? xs[k].b : xs[k].a; q[k] = (cell & 1) ? ys[k].b : ys[k].a; }
    multi_pairing(&r->c[cell], p, q, n);
  }
  free(p);
  free(q);
}
static void comt_add(comt* r, const comt* a, const comt* b) { for (int i = 0; i < 4; i++) f12_mul(&r->c[i], &a->c[i], &b->c[i]); }

/* --------------------------------------------------------- commit -------- */
enum { PPE = 0, MSMEG1 = 1, MSMEG2 = 2, QUAD = 3 };
static int xg(int ty) { return ty == PPE || ty == MSMEG1; }
static int yg(int ty) { return ty == PPE || ty == MSMEG2; }

static void commit_g1(com1* out, const g1a* xs, const fr* R, int m, const crs_t* k) { /* commit.rs:78-100 */
  com1* ru = (com1*)malloc(sizeof(com1) * m);
  com1_left_mul(ru, R, m, 2, k->u);
  for (int i = 0; i < m; i++) { com1 l; lin1(&l, &xs[i]); com1_add(&out[i], &l, &ru[i]); }
  free(ru);
}
static void commit_g2(com2* out, const g2a* ys, const fr* S, int n, const crs_t* k) {
  com2* sv = (com2*)malloc(sizeof(com2) * n);
  com2_left_mul(sv, S, n, 2, k->v);
  for (int i = 0; i < n; i++) { com2 l; lin2(&l, &ys[i]); com2_add(&out[i], &l, &sv[i]); }
  free(sv);
}
static void commit_fr1(com1* out, const fr* xs, const fr* r, int m, const crs_t* k) { /* commit.rs:125-156 */
  for (int i = 0; i < m; i++) { com1 s, t; slin1(&s, &xs[i], k); com1_smul(&t, &k->u[0], &r[i]); com1_add(&out[i], &s, &t); }
}
static void commit_fr2(com2* out, const fr* ys, const fr* s_, int n, const crs_t* k) {
  for (int i = 0; i < n; i++) { com2 s, t; slin2(&s, &ys[i], k); com2_smul(&t, &k->v[0], &s_[i]); com2_add(&out[i], &s, &t); }
}

/* ---------------------------------------------------------- prove -------- */
/* X/A are g1a arrays when xg(ty) else fr arrays; Y/B are g2a when yg(ty) else fr.  R: m x kx, S: n x ky, T: ky x kx */
static void prove(int ty, int m, int n, const void* X, const void* Y, const void* A, const void* B, const fr* G,
                  const fr* R, const fr* S, const fr* T, const crs_t* k, com2* pi, com1* theta) {
  int kx = xg(ty) ? 2 : 1, ky = yg(ty) ? 2 : 1;
  fr* Rt = (fr*)malloc(sizeof(fr) * m * kx); fr_transpose(Rt, R, m, kx);   /* kx x m */
  fr* St = (fr*)malloc(sizeof(fr) * n * ky); fr_transpose(St, S, n, ky);   /* ky x n */
  com2* mb = (com2*)malloc(sizeof(com2) * m);
  com2* my = (com2*)malloc(sizeof(com2) * n);
  com1* ma = (com1*)malloc(sizeof(com1) * n);
  com1* mx = (com1*)malloc(sizeof(com1) * m);
  for (int i = 0; i < m; i++) { if (yg(ty)) lin2(&mb[i], &((const g2a*)B)[i]); else slin2(&mb[i], &((const fr*)B)[i], k); }
  for (int j = 0; j < n; j++) { if (yg(ty)) lin2(&my[j], &((const g2a*)Y)[j]); else slin2(&my[j], &((const fr*)Y)[j], k); }
  for (int j = 0; j < n; j++) { if (xg(ty)) lin1(&ma[j], &((const g1a*)A)[j]); else slin1(&ma[j], &((const fr*)A)[j], k); }
  for (int i = 0; i < m; i++) { if (xg(ty)) lin1(&mx[i], &((const g1a*)X)[i]); else slin1(&mx[i], &((const fr*)X)[i], k); }
  /* pi */
  com2 x_rand_lin_b[2], x_rand_stmt_lin_y[2], pf_rand_stmt_com2[2];
  com2_left_mul(x_rand_lin_b, Rt, kx, m, mb);
  fr* x_rand_stmt = (fr*)malloc(sizeof(fr) * kx * n);
  fr_matmul(x_rand_stmt, Rt, kx, m, G, n);
  com2_left_mul(x_rand_stmt_lin_y, x_rand_stmt, kx, n, my);
  fr* tmp = (fr*)malloc(sizeof(fr) * kx * n);
  fr_matmul(tmp, Rt, kx, m, G, n); /* recomputed, as the reference does (prove.rs:139-140) */
  fr pf_rand_stmt[4], Tt[4];
  fr_matmul(pf_rand_stmt, tmp, kx, n, S, ky);
  fr_transpose(Tt, T, ky, kx); /* kx x ky */
  for (int i = 0; i < kx * ky; i++) { fr nt; fr_neg(&nt, &Tt[i]); fr_add(&pf_rand_stmt[i], &pf_rand_stmt[i], &nt); }
  com2_left_mul(pf_rand_stmt_com2, pf_rand_stmt, kx, ky, k->v); /* ky = 1 -> only v[0] */
  for (int i = 0; i < kx; i++) { com2 t; com2_add(&t, &x_rand_lin_b[i], &x_rand_stmt_lin_y[i]); com2_add(&pi[i], &t, &pf_rand_stmt_com2[i]); }
  /* theta */
  com1 y_rand_lin_a[2], y_rand_stmt_lin_x[2], pf_rand_com1[2];
  com1_left_mul(y_rand_lin_a, St, ky, n, ma);
  fr* Gt = (fr*)malloc(sizeof(fr) * m * n); fr_transpose(Gt, G, m, n); /* n x m */
  fr* y_rand_stmt = (fr*)malloc(sizeof(fr) * ky * m);
  fr_matmul(y_rand_stmt, St, ky, n, Gt, m);
  com1_left_mul(y_rand_stmt_lin_x, y_rand_stmt, ky, m, mx);
  com1_left_mul(pf_rand_com1, T, ky, kx, k->u); /* kx = 1 -> only u[0] */
  for (int i = 0; i < ky; i++) { com1 t; com1_add(&t, &y_rand_lin_a[i], &y_rand_stmt_lin_x[i]); com1_add(&theta[i], &t, &pf_rand_com1[i]); }
  free(Rt); free(St); free(mb); free(my); free(ma); free(mx); free(x_rand_stmt); free(tmp); free(Gt); free(y_rand_stmt);
}

/* ---------------------------------------------------------- verify ------- */
static int verify(int ty, int m, int n, const void* A, const void* B, const fr* G, const void* target,
                  const com1* xc, const com2* yc, const com2* pi, const com1* theta, const crs_t* k) {
  int kx = xg(ty) ? 2 : 1, ky = yg(ty) ? 2 : 1;
  com1* ma = (com1*)malloc(sizeof(com1) * n);
  com2* mb = (com2*)malloc(sizeof(com2) * m);
  for (int j = 0; j < n; j++) { if (xg(ty)) lin1(&ma[j], &((const g1a*)A)[j]); else slin1(&ma[j], &((const fr*)A)[j], k); }
  for (int i = 0; i < m; i++) { if (yg(ty)) lin2(&mb[i], &((const g2a*)B)[i]); else slin2(&mb[i], &((const fr*)B)[i], k); }
  comt lin_a_com_y, com_x_lin_b, com_x_stmt_com_y, lin_t, com1_pf2, pf1_com2, lhs, rhs, t;
  comt_pairing_sum(&lin_a_com_y, ma, yc, n);
  comt_pairing_sum(&com_x_lin_b, xc, mb, m);
  com2* stmt_com_y = (com2*)malloc(sizeof(com2) * m);
  com2_left_mul(stmt_com_y, G, m, n, yc); /* Gamma * d on the G2 side (verifier.rs:39-40) */
  comt_pairing_sum(&com_x_stmt_com_y, xc, stmt_com_y, m);
  fr one;
  memcpy(one.l, Q_ONE, 32);
  if (ty == PPE) {
    for (int i = 0; i < 3; i++) f12_one(&lin_t.c[i]);
    lin_t.c[3] = *(const fp12*)target;
  } else if (ty == MSMEG1) {
    com1 a; com2 b; lin1(&a, (const g1a*)target); slin2(&b, &one, k); comt_pairing(&lin_t, &a, &b);
  } else if (ty == MSMEG2) {
    com1 a; com2 b; slin1(&a, &one, k); lin2(&b, (const g2a*)target); comt_pairing(&lin_t, &a, &b);
  } else {
    com1 a; com2 b, bt; slin1(&a, &one, k); slin2(&b, &one, k); com2_smul(&bt, &b, (const fr*)target); comt_pairing(&lin_t, &a, &bt);
  }
  if (kx == 2) comt_pairing_sum(&com1_pf2, k->u, pi, 2); else comt_pairing(&com1_pf2, &k->u[0], &pi[0]);
  if (ky == 2) comt_pairing_sum(&pf1_com2, theta, k->v, 2); else comt_pairing(&pf1_com2, &theta[0], &k->v[0]);
  comt_add(&t, &lin_a_com_y, &com_x_lin_b); comt_add(&lhs, &t, &com_x_stmt_com_y);
  comt_add(&t, &lin_t, &com1_pf2); comt_add(&rhs, &t, &pf1_com2);
  int ok = 1;
  for (int i = 0; i < 4; i++) ok &= f12_eq(&lhs.c[i], &rhs.c[i]);
  free(ma); free(mb); free(stmt_com_y);
  return ok;
}

/* ------------------------------------------------------- exported API ---- */
/* all arrays use the boundary layout of include/gs_amd.h (Montgomery limbs) */
int ref_sizes(int* out) { out[0] = sizeof(fp); out[1] = sizeof(fr); out[2] = sizeof(g1a); out[3] = sizeof(g2a); out[4] = sizeof(fp12); out[5] = sizeof(crs_t); return 0; }
void ref_g1_mul(const void* p, const void* k, void* out) { g1_mul((g1a*)out, (const g1a*)p, (const fr*)k); }
void ref_g2_mul(const void* p, const void* k, void* out) { g2_mul((g2a*)out, (const g2a*)p, (const fr*)k); }
void ref_multi_pairing(int n, const void* ps, const void* qs, void* out) { multi_pairing((fp12*)out, (const g1a*)ps, (const g2a*)qs, n); }
void ref_pairing_sum(int n, const void* xs, const void* ys, void* out) { comt_pairing_sum((comt*)out, (const com1*)xs, (const com2*)ys, n); }
void ref_left_mul_com1(int rows, int cols, const void* lhs, const void* col, void* out) { com1_left_mul((com1*)out, (const fr*)lhs, rows, cols, (const com1*)col); }
void ref_left_mul_com2(int rows, int cols, const void* lhs, const void* col, void* out) { com2_left_mul((com2*)out, (const fr*)lhs, rows, cols, (const com2*)col); }
void ref_gt_pow(const void* base, const void* k_mont, void* out) {
  u64 k[4];
  fr_from_mont(k, (const fr*)k_mont);
  fp12 acc, b = *(const fp12*)base;
  f12_one(&acc);
  for (int i = 255; i >= 0; i--) { f12_sqr(&acc, &acc); if ((k[i >> 6] >> (i & 63)) & 1) f12_mul(&acc, &acc, &b); }
  *(fp12*)out = acc;
}
void ref_commit_and_prove(int ty, int m, int n, const void* X, const void* Y, const void* A, const void* B,
                          const void* G, const void* R, const void* S, const void* T, const void* crs, void* xc,
                          void* yc, void* pi, void* theta) {
  const crs_t* k = (const crs_t*)crs;
  if (xc) { if (xg(ty)) commit_g1((com1*)xc, (const g1a*)X, (const fr*)R, m, k); else commit_fr1((com1*)xc, (const fr*)X, (const fr*)R, m, k); }
  if (yc) { if (yg(ty)) commit_g2((com2*)yc, (const g2a*)Y, (const fr*)S, n, k); else commit_fr2((com2*)yc, (const fr*)Y, (const fr*)S, n, k); }
  prove(ty, m, n, X, Y, A, B, (const fr*)G, (const fr*)R, (const fr*)S, (const fr*)T, k, (com2*)pi, (com1*)theta);
}
int ref_verify(int ty, int m, int n, const void* A, const void* B, const void* G, const void* target, const void* xc,
               const void* yc, const void* pi, const void* theta, const void* crs) {
  return verify(ty, m, n, A, B, (const fr*)G, target, (const com1*)xc, (const com2*)yc, (const com2*)pi,
                (const com1*)theta, (const crs_t*)crs);
}
/* Matrix<Fr>::right_mul (data_structures.rs:824-869): out (ar x bc) = a (ar x ac) * b (ac x bc), Montgomery Fr */
void ref_fr_matmul(int ar, int ac, int bc, const void* a, const void* b, void* out) { fr_matmul((fr*)out, (const fr*)a, ar, ac, (const fr*)b, bc); }
u64 ref_fpmul_count(int reset) { u64 v = g_fpmul_count; if (reset) g_fpmul_count = 0; return v; }

/* ------------------------------------------------ self-contained bench --- */
static u64 sm_state;
static u64 sm_next(void) {
  u64 z = (sm_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void rand_fr(fr* r) { /* uniform limbs taken as the Montgomery representation */
  for (;;) {
    for (int i = 0; i < 4; i++) r->l[i] = sm_next();
    r->l[3] &= (Q_MOD[3] | (Q_MOD[3] >> 1) | (Q_MOD[3] >> 2) | (Q_MOD[3] >> 4) | (Q_MOD[3] >> 8) | (Q_MOD[3] >> 16) | (Q_MOD[3] >> 32));
    u64 br = 0;
    for (int j = 0; j < 4; j++) { u128 s = (u128)r->l[j] - Q_MOD[j] - br; br = (u64)(s >> 127); }
    if (br) return;
  }
}
typedef struct {
  int m, n;
  crs_t crs;
  g1a *X, *A; g2a *Y, *B; fr *G, *R, *S, *T; fp12* target;
  com1* xc; com2* yc; com2* pi; com1* th;
  int* ok;
  int units, next, nthreads;
  pthread_mutex_t mu;
  u64 fpmuls;
} bench_t;
static void* bench_worker(void* arg) {
  bench_t* b = (bench_t*)arg;
  int m = b->m, n = b->n;
  g_fpmul_count = 0;
  for (;;) {
    pthread_mutex_lock(&b->mu);
    int e = b->next++;
    pthread_mutex_unlock(&b->mu);
    if (e >= b->units) break;
    ref_commit_and_prove(PPE, m, n, b->X + e * m, b->Y + e * n, b->A + e * n, b->B + e * m, b->G + e * m * n,
                         b->R + e * m * 2, b->S + e * n * 2, b->T + e * 4, &b->crs, b->xc + e * m, b->yc + e * n,
                         b->pi + e * 2, b->th + e * 2);
    b->ok[e] = verify(PPE, m, n, b->A + e * n, b->B + e * m, b->G + e * m * n, &b->target[e], b->xc + e * m,
                      b->yc + e * n, b->pi + e * 2, b->th + e * 2, &b->crs);
  }
  pthread_mutex_lock(&b->mu);
  b->fpmuls += g_fpmul_count;
  pthread_mutex_unlock(&b->mu);
  return 0;
}
/* Generates `units` satisfied PPE m x n instances (seeded), runs commit_and_prove + verify of each on
 * `threads` threads over equations.  Returns seconds of the timed region; *all_ok, *fpmuls out. */
double ref_bench_ppe(int units, int m, int n, int threads, u64 seed, int* all_ok, u64* fpmuls) {
  bench_t b;
  memset(&b, 0, sizeof b);
  b.m = m; b.n = n; b.units = units; b.nthreads = threads;
  sm_state = seed;
  g1a g1s; g2a g2s;
  memcpy(&g1s, G1_GEN, sizeof g1s);
  memcpy(&g2s, G2_GEN, sizeof g2s);
  fr al, be, a1, a2, t1, t2;
  rand_fr(&al); rand_fr(&be); rand_fr(&a1); rand_fr(&a2); rand_fr(&t1); rand_fr(&t2);
  g1a p1, q1, u1, v1; g2a p2, q2, u2, v2;
  g1_mul(&p1, &g1s, &al); g2_mul(&p2, &g2s, &be);
  g1_mul(&q1, &p1, &a1); g2_mul(&q2, &p2, &a2);
  g1_mul(&u1, &p1, &t1); g2_mul(&u2, &p2, &t2);
  g1_mul(&v1, &q1, &t1); g2_mul(&v2, &q2, &t2);
  b.crs.u[0].a = p1; b.crs.u[0].b = q1; b.crs.u[1].a = u1; b.crs.u[1].b = v1;
  b.crs.v[0].a = p2; b.crs.v[0].b = q2; b.crs.v[1].a = u2; b.crs.v[1].b = v2;
  b.crs.g1 = p1; b.crs.g2 = p2;
  multi_pairing(&b.crs.gt, &p1, &p2, 1);
  b.X = malloc(sizeof(g1a) * units * m); b.A = malloc(sizeof(g1a) * units * n);
  b.Y = malloc(sizeof(g2a) * units * n); b.B = malloc(sizeof(g2a) * units * m);
  b.G = malloc(sizeof(fr) * units * m * n); b.R = malloc(sizeof(fr) * units * m * 2);
  b.S = malloc(sizeof(fr) * units * n * 2); b.T = malloc(sizeof(fr) * units * 4);
  b.target = malloc(sizeof(fp12) * units);
  b.xc = malloc(sizeof(com1) * units * m); b.yc = malloc(sizeof(com2) * units * n);
  b.pi = malloc(sizeof(com2) * units * 2); b.th = malloc(sizeof(com1) * units * 2);
  b.ok = calloc(units, sizeof(int));
  for (int e = 0; e < units; e++) {
    fr xs[64], ys[64], as[64], bs[64], s, t;
    memset(&s, 0, sizeof s);
    for (int i = 0; i < m; i++) { rand_fr(&xs[i]); g1_mul(&b.X[e * m + i], &p1, &xs[i]); }
    for (int j = 0; j < n; j++) { rand_fr(&ys[j]); g2_mul(&b.Y[e * n + j], &p2, &ys[j]); }
    for (int j = 0; j < n; j++) { rand_fr(&as[j]); g1_mul(&b.A[e * n + j], &p1, &as[j]); fr_mul(&t, &as[j], &ys[j]); fr_add(&s, &s, &t); }
    for (int i = 0; i < m; i++) { rand_fr(&bs[i]); g2_mul(&b.B[e * m + i], &p2, &bs[i]); fr_mul(&t, &xs[i], &bs[i]); fr_add(&s, &s, &t); }
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) {
      fr* g = &b.G[(e * m + i) * n + j];
      rand_fr(g);
      fr_mul(&t, &xs[i], g); fr_mul(&t, &t, &ys[j]); fr_add(&s, &s, &t);
    }
    for (int i = 0; i < m * 2; i++) rand_fr(&b.R[e * m * 2 + i]);
    for (int i = 0; i < n * 2; i++) rand_fr(&b.S[e * n * 2 + i]);
    for (int i = 0; i < 4; i++) rand_fr(&b.T[e * 4 + i]);
    ref_gt_pow(&b.crs.gt, &s, &b.target[e]);
  }
  pthread_mutex_init(&b.mu, 0);
  struct timespec t0, t1_;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_t* th = malloc(sizeof(pthread_t) * threads);
  for (int i = 0; i < threads; i++) pthread_create(&th[i], 0, bench_worker, &b);
  for (int i = 0; i < threads; i++) pthread_join(th[i], 0);
  clock_gettime(CLOCK_MONOTONIC, &t1_);
  int ok = 1;
  for (int e = 0; e < units; e++) ok &= b.ok[e];
  *all_ok = ok;
  *fpmuls = b.fpmuls;
  free(b.X); free(b.A); free(b.Y); free(b.B); free(b.G); free(b.R); free(b.S); free(b.T); free(b.target);
  free(b.xc); free(b.yc); free(b.pi); free(b.th); free(b.ok); free(th);
  return (t1_.tv_sec - t0.tv_sec) + 1e-9 * (t1_.tv_nsec - t0.tv_nsec);
}
