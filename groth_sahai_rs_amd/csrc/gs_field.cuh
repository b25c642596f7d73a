// Montgomery prime-field arithmetic for gfx950 (one field element per lane,
// 32-bit limbs held in VGPRs as a clang ext-vector so that the out-of-line
// multiply takes and returns its operands in registers).
//
// Replaces arkworks' Fp<MontBackend<_,N>,N> (ark-ff ^0.5; external to the
// reference, call sites src/data_structures.rs:780-907 for Fr and everything
// below Com1/Com2::scalar_mul :336-342,381-387 for Fq).  In-memory form is
// identical to arkworks': little-endian limbs of a * 2^(32N) mod p.
//
// The same source compiles for the host (clang++, no HIP) as the "CPU twin"
// used only by tests/ to validate the algorithms without a GPU.
#pragma once
#include <stdint.h>

// GS_FN_ATTR: optional attribute for every out-of-line device function (experiments: e.g.
// -DGS_FN_ATTR='__attribute__((amdgpu_waves_per_eu(2,2)))' builds them for two waves per SIMD)
#ifndef GS_FN_ATTR
#define GS_FN_ATTR
#endif
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GS_HD __host__ __device__ __forceinline__
#define GS_HD_NOINLINE __host__ __device__ __attribute__((noinline)) GS_FN_ATTR
#else
#define GS_HD inline __attribute__((always_inline))
#define GS_HD_NOINLINE __attribute__((noinline))
#endif

#include "gs_montmul_asm.h"

namespace gs {

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x12 __attribute__((ext_vector_type(12)));
template <int N> struct LimbVec;
template <> struct LimbVec<8> { typedef u32x8 type; };
template <> struct LimbVec<12> { typedef u32x12 type; };

// Modulus descriptors: Fq (base field) and Fr (scalar field) of curve C.
template <class C> struct FqM {
  static constexpr int N = C::N;
  static constexpr int BITS = C::P_BITS;
  static constexpr uint32_t INV = C::P_INV;
  GS_HD static uint32_t mod(int i) { return C::P[i]; }
  GS_HD static uint32_t one(int i) { return C::P_ONE[i]; }
  GS_HD static uint32_t r2(int i) { return C::P_R2[i]; }
};
template <class C> struct FrM {
  typedef C Curve;
  static constexpr int N = C::NR;
  static constexpr int BITS = C::Q_BITS;
  static constexpr uint32_t INV = C::Q_INV;
  GS_HD static uint32_t mod(int i) { return C::Q[i]; }
  GS_HD static uint32_t one(int i) { return C::Q_ONE[i]; }
  GS_HD static uint32_t r2(int i) { return C::Q_R2[i]; }
};

// Storage is a plain limb array (tight 4*N bytes, memcpy-compatible with the
// boundary layout); the ext-vector form (padded to 32/64 bytes) exists only in
// registers across the out-of-line multiply's call boundary.
template <class M> struct Fe {
  typedef typename LimbVec<M::N>::type V;
  uint32_t v[M::N];
};
template <class M> GS_HD typename Fe<M>::V to_vec(const Fe<M>& a) {
  typename Fe<M>::V r;
#pragma unroll
  for (int j = 0; j < M::N; j++) r[j] = a.v[j];
  return r;
}
template <class M> GS_HD Fe<M> from_vec(typename Fe<M>::V a) {
  Fe<M> r;
#pragma unroll
  for (int j = 0; j < M::N; j++) r.v[j] = a[j];
  return r;
}

// ---------------------------------------------------------------------------
// raw limb-vector kernels
// ---------------------------------------------------------------------------

// Montgomery product, CIOS with the "no-carry" simplification (top modulus bit
// is clear for both curves), fully unrolled: 2*N*N v_mad_u64_u32 + N v_mul_lo.
template <class M>
GS_HD_NOINLINE typename Fe<M>::V mont_mul_raw(typename Fe<M>::V a, typename Fe<M>::V b) {
  constexpr int N = M::N;
  uint32_t t[N + 1];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NO_ASM)
  {
    // hand-scheduled product-scanning kernel (gs_montmul_asm.h): 2 VALU ops per MAC
    uint32_t av[N], bv[N], rv[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      av[i] = a[i];
      bv[i] = b[i];
    }
    if constexpr (N == 12)
      mont_mul_asm_12<M>(rv, av, bv);
    else
      mont_mul_asm_8<M>(rv, av, bv);
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = rv[i];
    t[N] = 0;
  }
#else
#pragma unroll
  for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      c += (uint64_t)a[j] * b[i] + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    uint32_t tn = t[N] + (uint32_t)c;  // cannot overflow (no-carry property)
    uint32_t m = t[0] * M::INV;
    c = (uint64_t)m * M::mod(0) + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < N; j++) {
      c += (uint64_t)m * M::mod(j) + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += tn;
    t[N - 1] = (uint32_t)c;
    t[N] = (uint32_t)(c >> 32);
  }
#endif
  // result < 2p: one conditional subtraction
  uint32_t d[N];
  uint32_t br = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    uint64_t s = (uint64_t)t[j] - M::mod(j) - br;
    d[j] = (uint32_t)s;
    br = (uint32_t)(s >> 63);
  }
  bool keep = (t[N] == 0) && br;  // t < p
  typename Fe<M>::V r;
#pragma unroll
  for (int j = 0; j < N; j++) r[j] = keep ? t[j] : d[j];
  return r;
}

template <class M> GS_HD Fe<M> mul(const Fe<M>& a, const Fe<M>& b) {
  return from_vec<M>(mont_mul_raw<M>(to_vec(a), to_vec(b)));
}
template <class M> GS_HD Fe<M> sqr(const Fe<M>& a) { return mul(a, a); }

template <class M> GS_HD Fe<M> add(const Fe<M>& a, const Fe<M>& b) {
  constexpr int N = M::N;
  uint32_t s[N], d[N];
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    uint64_t x = (uint64_t)a.v[j] + b.v[j] + c;
    s[j] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
  uint32_t br = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    uint64_t x = (uint64_t)s[j] - M::mod(j) - br;
    d[j] = (uint32_t)x;
    br = (uint32_t)(x >> 63);
  }
  bool keep = br && !c;  // s < p
  Fe<M> r;
#pragma unroll
  for (int j = 0; j < N; j++) r.v[j] = keep ? s[j] : d[j];
  return r;
}

template <class M> GS_HD Fe<M> sub(const Fe<M>& a, const Fe<M>& b) {
  constexpr int N = M::N;
  uint32_t d[N];
  uint32_t br = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    uint64_t x = (uint64_t)a.v[j] - b.v[j] - br;
    d[j] = (uint32_t)x;
    br = (uint32_t)(x >> 63);
  }
  uint32_t mask = 0u - br;
  uint32_t c = 0;
  Fe<M> r;
#pragma unroll
  for (int j = 0; j < N; j++) {
    uint64_t x = (uint64_t)d[j] + (M::mod(j) & mask) + c;
    r.v[j] = (uint32_t)x;
    c = (uint32_t)(x >> 32);
  }
  return r;
}

template <class M> GS_HD Fe<M> fzero() {
  Fe<M> r;
#pragma unroll
  for (int j = 0; j < M::N; j++) r.v[j] = 0;
  return r;
}
template <class M> GS_HD Fe<M> fone() {
  Fe<M> r;
#pragma unroll
  for (int j = 0; j < M::N; j++) r.v[j] = M::one(j);
  return r;
}
template <class M> GS_HD bool is_zero(const Fe<M>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < M::N; j++) o |= a.v[j];
  return o == 0;
}
template <class M> GS_HD bool eq(const Fe<M>& a, const Fe<M>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < M::N; j++) o |= a.v[j] ^ b.v[j];
  return o == 0;
}
template <class M> GS_HD Fe<M> neg(const Fe<M>& a) {
  if (is_zero(a)) return a;
  Fe<M> p;
#pragma unroll
  for (int j = 0; j < M::N; j++) p.v[j] = M::mod(j);
  return sub(p, a);  // p - a, no borrow
}
template <class M> GS_HD Fe<M> dbl(const Fe<M>& a) { return add(a, a); }
template <class M> GS_HD Fe<M> select(bool c, const Fe<M>& a, const Fe<M>& b) {  // c ? a : b
  Fe<M> r;
#pragma unroll
  for (int j = 0; j < M::N; j++) r.v[j] = c ? a.v[j] : b.v[j];
  return r;
}
// Montgomery -> canonical integer limbs (multiply by 1)
template <class M> GS_HD Fe<M> from_mont(const Fe<M>& a) {
  Fe<M> o = fzero<M>();
  o.v[0] = 1;
  return mul(a, o);
}
template <class M> GS_HD Fe<M> to_mont(const Fe<M>& a) {
  Fe<M> r2;
#pragma unroll
  for (int j = 0; j < M::N; j++) r2.v[j] = M::r2(j);
  return mul(a, r2);
}
// small-constant multiply by repeated addition (k <= 16)
template <class M> GS_HD Fe<M> mul_small(const Fe<M>& a, int k) {
  Fe<M> r = fzero<M>(), t = a;
  while (k) {
    if (k & 1) r = add(r, t);
    t = dbl(t);
    k >>= 1;
  }
  return r;
}

// a^(mod-2): Fermat inversion, 4-bit fixed window over the constant exponent.
// inv(0) = 0.
template <class M> GS_HD_NOINLINE typename Fe<M>::V inv_raw(typename Fe<M>::V av) {
  constexpr int N = M::N;
  Fe<M> a = from_vec<M>(av);
  Fe<M> tab[16];
  tab[0] = fone<M>();
  tab[1] = a;
  for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
  // exponent = mod - 2 (mod is odd and > 2, so only limb 0 changes, no borrow)
  Fe<M> r = fone<M>();
  bool started = false;
  for (int w = N * 8 - 1; w >= 0; w--) {
    int limb = w >> 3, sh = (w & 7) * 4;
    uint32_t e = M::mod(limb);
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
  return to_vec(r);
}
template <class M> GS_HD Fe<M> inv(const Fe<M>& a) { return from_vec<M>(inv_raw<M>(to_vec(a))); }

// bit i of a canonical (non-Montgomery) element
template <class M> GS_HD uint32_t get_bit(const Fe<M>& a, int i) { return (a.v[i >> 5] >> (i & 31)) & 1u; }
// w-bit window starting at bit i (may straddle a limb; bits past the top read 0)
template <class M> GS_HD uint32_t get_bits(const Fe<M>& a, int i, int w) {
  int limb = i >> 5, sh = i & 31;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < M::N; j++) {
    if (j == limb) lo = a.v[j];
    if (j == limb + 1) hi = a.v[j];
  }
  uint64_t x = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(x >> sh) & ((1u << w) - 1u);
}

}  // namespace gs
