// Device kernels of the Groth-Sahai engine.  One lane = one task; a task is a
// descriptor-driven unit of group arithmetic, so that the four equation types
// of the reference (src/statement.rs:117-192) and both groups share the same
// handful of kernels.  Per batch the host (gs_amd.hip) uploads small per-shape
// task tables; per-equation data is addressed as  base[arr] + eq*stride[arr].
//
//   k_prep_*   Fr mini-GEMMs  Psi = R^T Gamma, Omega = Psi S - T^T, Phi = S^T Gamma^T
//              (prove.rs:133,139-142,154) -> canonical scalars in the "pool"
//   k_var      one variable-base scalar multiplication per lane (Jacobian out)
//   k_fix      <= 2 fixed-base terms from the CRS window tables (+ affine addend)
//   k_red      sum the partial slots of a commitment-group element, ONE shared
//              inversion per element (Montgomery trick), normalised affine out
//   k_miller   multi-Miller loop over <= 3 pairs  -> un-exponentiated Fp12
//   k_final    product of a cell's partial Miller values, final exponentiation,
//              compare (verifier.rs:50-53)
#pragma once
#include <type_traits>
#include <utility>

#include "gs_pairing.cuh"
#include "gs_coop.cuh"
#include "gs_wire.cuh"

namespace gs {

#ifndef GS_WPE
#define GS_WPE 1
#endif
constexpr int MAX_ARR = 8;
struct ArrTab {
  const uint8_t* base[MAX_ARR];
  uint32_t stride[MAX_ARR];  // bytes per equation; 0 = shared by the whole batch
};
struct OutTab {
  uint8_t* base[4];
  uint32_t stride[4];
};
struct VarTask {
  uint32_t s_idx, p_idx, slot;  // 32-bit: large-arity equations (m, n ~ 334, benches/bench.rs:451-498) have > 2^16 scalars
  uint8_t p_arr, neg, pad0, pad1;  // neg: use -P
};
constexpr int GRP_OUT = 4;  // outputs one Straus lane can serve from one table build
struct GrpTask {  // one lane of k_var_multi: `no` outputs over the SAME nt bases (in the same order, same signs);
  uint32_t nt, no;             // output o sums VarTasks first[o] .. first[o] + nt - 1 into slot[o]
  uint32_t first[GRP_OUT], slot[GRP_OUT];
};
struct FixTask {
  uint32_t s0, s1, a_idx, slot;
  uint8_t t0, t1, a_arr, a_neg;  // 0xFF = absent
};
struct RedTask {
  uint32_t b0, e0, b1, e1, out_idx;
  uint8_t out_arr, pad, pad1, pad2;
};
// one G2 argument with its TWO G1 partners: P(a) = parr[p_arr][p_idx + a], a = 0, 1
struct PairRef {
  uint8_t p_arr, q_arr, neg, pad;
  uint32_t p_idx, q_idx;
};
#ifndef GS_MILLER_CAP
#define GS_MILLER_CAP 12
#endif
constexpr int MILLER_CH = GS_MILLER_CAP;  // capacity of a Miller lane; the host picks the chunk per batch size
struct MillerTask {
  uint8_t np, b, single, pad1;  // single: only the P(0) partner is used (one accumulator)
  PairRef pr[MILLER_CH];
};
// where the Miller partials of cell c = 2a + b live: tasks [lo, hi), slot sub (0/1) of each task
struct CellMap {
  int lo[4], hi[4], sub[4];
};
struct PoolMap {  // offsets (in scalars) into the per-equation pool
  int RC, SC, PSI, PHI, OM, TC, RHO, SIG, XC, YC, GC, AC, BC, NT, RH, AR, NR, total;
};

template <class T> __device__ __forceinline__ T ld(const uint8_t* p) { return *reinterpret_cast<const T*>(p); }

// ---- segmented launches ---------------------------------------------------------------------------------------------
// The kernels of the prove / verify path are written as BODIES: `struct k_xxx { static __device__ void run(size_t g,
// args...); }`, g = the lane's index within its launch.  One generic __global__ function, k_seg<Body, args...>, runs a
// body over up to MAX_SEG SEGMENTS: consecutive, block-aligned lane ranges with their own argument packs.  An ordinary
// launch is one segment.  A mixed batch (gs_prove_mixed / gs_verify_mixed: sub-batches of different equation types and
// shapes, each with its own task tables and arrays) MERGES the launches of its parts that run the same body into one
// launch with a segment per part, so that a few thousand equations of three types fill the chip like a homogeneous
// batch of their total size -- dispatches from several streams do not share these 512-register kernels' SIMDs
// (profiles/r3/mixed_streams.txt).  The segment of a lane is wave-uniform (segments start at block boundaries, blocks
// are one wave), so its argument pack is read with scalar loads from the kernel-argument segment exactly as the
// arguments of an ordinary kernel are.
constexpr int MAX_SEG = 4;
template <size_t I, class T> struct PackLeaf {
  T v;
};
template <class Seq, class... A> struct PackImpl;
template <size_t... I, class... A> struct PackImpl<std::index_sequence<I...>, A...> : PackLeaf<I, A>... {};
template <class... A> using Pack = PackImpl<std::index_sequence_for<A...>, A...>;
template <size_t I, class T> __host__ __device__ __forceinline__ const T& pack_get(const PackLeaf<I, T>& l) { return l.v; }
template <class... A> struct Segs {
  int n, pad;
  // clock stamps (kernel profile only, else null): every wave adds its own d s_memtime (shader cycles), d s_memrealtime
  // (100 MHz) and 1 to stamp[0..2], so that the host reads the clock THIS launch ran at -- sum d memtime / sum d
  // memrealtime x 100 MHz -- from the launch it has just timed (bench.py: the ALU peak at the kernel's own clock)
  unsigned long long* stamp;
  size_t lo[MAX_SEG];  // first lane of each segment (multiples of the block size, ascending; lo[0] = 0)
  Pack<A...> a[MAX_SEG];
};
template <class Body, class... A, size_t... I>
__device__ __forceinline__ void seg_call(size_t g, const Pack<A...>& p, std::index_sequence<I...>) {
  Body::run(g, pack_get<I>(p)...);
}
#ifndef GS_KSEG_ATTR
#define GS_KSEG_ATTR
#endif
// waves per SIMD a body is BUILT for (the second argument of __launch_bounds__ = the minimum the compiler must allow:
// 2 caps the kernel at 256 registers).  Default 1: every heavy kernel owns the whole register file of its SIMD.
template <class B, class = void> struct BodyWpe {
  static constexpr int v = GS_WPE;
};
template <class B> struct BodyWpe<B, std::void_t<decltype(B::WPE)>> {
  static constexpr int v = B::WPE;
};
template <class Body, class... A> __global__ void __launch_bounds__(64, BodyWpe<Body>::v) GS_KSEG_ATTR k_seg(Segs<A...> S) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int s = 0;
  for (int i = 1; i < S.n; i++)
    if (g >= S.lo[i]) s = i;
  s = __builtin_amdgcn_readfirstlane(s);
  unsigned long long t0 = 0, r0 = 0;
  if (S.stamp) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  seg_call<Body>(g - S.lo[s], S.a[s], std::index_sequence_for<A...>{});
  if (S.stamp) {  // (a body's early `return` ends the inlined body, not the kernel: every wave gets here)
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
      atomicAdd(S.stamp, t1 - t0);
      atomicAdd(S.stamp + 1, r1 - r0);
      atomicAdd(S.stamp + 2, 1ull);
    }
  }
}

// ---- boundary <-> internal I/O --------------------------------------------------
// API arrays hold arkworks' saturated Montgomery limbs (BFq); everything the
// kernels keep between launches (window tables, Jacobian partials, Miller partials)
// stays in the internal radix-2^28 form.
template <class C> constexpr size_t aff_bytes(const Aff<Fq<C>>*) { return 2 * sizeof(BFq<C>); }
template <class C> constexpr size_t aff_bytes(const Aff<Fp2<C>>*) { return 4 * sizeof(BFq<C>); }
template <class C> GS_HD void aff_load(Aff<Fq<C>>& r, const uint8_t* p) {
  const BFq<C>* b = reinterpret_cast<const BFq<C>*>(p);
  BFq<C> x = b[0], y = b[1];
  r.x = fq_from_boundary<C>(x.w);
  r.y = fq_from_boundary<C>(y.w);
}
template <class C> GS_HD void aff_load(Aff<Fp2<C>>& r, const uint8_t* p) {
  const BFq<C>* b = reinterpret_cast<const BFq<C>*>(p);
  BFq<C> t[4] = {b[0], b[1], b[2], b[3]};
  r.x = fp2_from_boundary<C>(t);
  r.y = fp2_from_boundary<C>(t + 2);
}
template <class C> GS_HD void aff_store(uint8_t* p, const Aff<Fq<C>>& a) {
  BFq<C>* b = reinterpret_cast<BFq<C>*>(p);
  BFq<C> x, y;
  fq_to_boundary<C>(x.w, a.x);
  fq_to_boundary<C>(y.w, a.y);
  b[0] = x;
  b[1] = y;
}
template <class C> GS_HD void aff_store(uint8_t* p, const Aff<Fp2<C>>& a) {
  BFq<C>* b = reinterpret_cast<BFq<C>*>(p);
  BFq<C> t[4];
  fp2_to_boundary<C>(t, a.x);
  fp2_to_boundary<C>(t + 2, a.y);
  b[0] = t[0];
  b[1] = t[1];
  b[2] = t[2];
  b[3] = t[3];
}
#define AFFB(C, F) (aff_bytes<C>((const Aff<F>*)nullptr))

// --------------------------------------------------------------------------
// generic helpers
// --------------------------------------------------------------------------
template <class C, class F, bool ENDO = true>
__global__ void __launch_bounds__(64, GS_WPE) k_smul_batch(size_t n, const uint8_t* p, int broadcast, const Fr<C>* k,
                                                   uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> P;
  aff_load<C>(P, p + (broadcast ? 0 : i) * AFFB(C, F));
  Jac<F> J;
  jac_smul_any<C, F, ENDO>(J, P, from_mont(k[i]));
  Aff<F> R;
  jac_to_aff(R, J);
  aff_store<C>(out + i * AFFB(C, F), R);
}

// pts[5] <- pts[3] + pts[5]  (W.1 = u1.1 + generator), single lane
template <class C, class F> __global__ void k_crs_derive(uint8_t* pts) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Aff<F> a, b;
  aff_load<C>(a, pts + 3 * AFFB(C, F));
  aff_load<C>(b, pts + 5 * AFFB(C, F));
  Jac<F> j;
  jac_from_aff(j, a);
  jac_madd(j, j, b);
  Aff<F> r;
  jac_to_aff(r, j);
  aff_store<C>(pts + 5 * AFFB(C, F), r);
}

// window tables: tab[(b*32 + w)*256 + d] = d * 2^(8w) * base[b]   (d = 0 -> identity)
template <class C, class F>
__global__ void __launch_bounds__(64, GS_WPE) k_build_tables(int nb, const uint8_t* bases, Aff<F>* tab) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nb * 32 * 256) return;
  int d = (int)(i & 255), w = (int)((i >> 8) & 31), b = (int)(i >> 13);
  Fr<C> k = fzero<FrM<C>>();
  k.v[w >> 2] = (uint32_t)d << ((w & 3) * 8);
  Aff<F> B;
  aff_load<C>(B, bases + b * AFFB(C, F));
  Jac<F> J;
  jac_smul(J, B, k);  // integer multiple d*2^(8w) may exceed r: plain path (endo decomposition assumes k < r)
  Aff<F> R;
  jac_to_aff(R, J);
  tab[i] = R;
}

// Second level: 16-bit windows, tab16[(b*16 + w)*65536 + d] = d * 2^(16w) * base[b], each entry the sum of two
// first-level entries (one mixed addition per entry, one inversion per 16 entries).  1.8 GB per CRS for both groups -- a
// fixed-base scalar then costs 16 mixed additions instead of 32.  288 GB of HBM is what makes this the right trade.
template <class C, class F>
__global__ void __launch_bounds__(64, GS_WPE) k_build_tables16(int nb, const Aff<F>* tab8, Aff<F>* tab16) {
  // one lane = 16 consecutive entries (same base, window and high byte), normalised with ONE inversion
  constexpr int G = 16;
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nb * 16 * 65536 / G) return;
  size_t i0 = g * G;
  uint32_t d0 = (uint32_t)(i0 & 65535u);
  int w = (int)((i0 >> 16) & 15), b = (int)(i0 >> 20);
  Aff<F> hi = tab8[((size_t)b * 32 + 2 * w + 1) * 256 + (d0 >> 8)];
  Jac<F> J[G];
  F pre[G];
  F acc = one_of<F>();
  for (int t = 0; t < G; t++) {
    Aff<F> lo = tab8[((size_t)b * 32 + 2 * w) * 256 + ((d0 + t) & 255u)];
    jac_from_aff(J[t], lo);
    jac_madd(J[t], J[t], hi);
    pre[t] = acc;
    if (!is_zero_limbs(J[t].z)) acc = mul(acc, J[t].z);
  }
  F inv_all = inv(acc);
  for (int t = G - 1; t >= 0; t--) {
    Aff<F> R;
    if (is_zero_limbs(J[t].z)) {
      R.x = zero_of<F>();
      R.y = zero_of<F>();
    } else {
      F zi = mul(inv_all, pre[t]);
      inv_all = mul(inv_all, J[t].z);
      jac_to_aff_zinv(R, J[t], zi);
    }
    tab16[i0 + t] = R;
  }
}

// --------------------------------------------------------------------------
// Fr preparation for prove (one lane per equation)
// --------------------------------------------------------------------------
template <class C>
struct k_prep_prove {
  static __device__ __forceinline__ void run(size_t e_in, size_t N, int m, int n, int kx, int ky, const Fr<C>* G, const Fr<C>* R, const Fr<C>* S,
                 const Fr<C>* T, const Fr<C>* xs, const Fr<C>* ys, const Fr<C>* as, const Fr<C>* bs, PoolMap pm,
                 Fr<C>* pool, int shared_vars) {
  size_t e = e_in;
  if (e >= N) return;
  typedef Fr<C> S_;
  // shared_vars: a Statement -- every equation of the batch is over the SAME variables (and commit randomness)
  const size_t ev = shared_vars ? 0 : e;
  G += e * m * n;
  R += ev * m * kx;
  S += ev * n * ky;
  T += e * ky * kx;
  if (xs) xs += ev * m;
  if (ys) ys += ev * n;
  if (as) as += e * n;
  if (bs) bs += e * m;
  S_* P = pool + e * pm.total;
  for (int i = 0; i < m * kx; i++) P[pm.RC + i] = from_mont(R[i]);
  for (int i = 0; i < n * ky; i++) P[pm.SC + i] = from_mont(S[i]);
  for (int i = 0; i < ky * kx; i++) P[pm.TC + i] = from_mont(T[i]);
  if (xs)
    for (int i = 0; i < m; i++) P[pm.XC + i] = from_mont(xs[i]);
  if (ys)
    for (int j = 0; j < n; j++) P[pm.YC + j] = from_mont(ys[j]);
  // Psi = R^T Gamma (kx x n); Omega = Psi S - T^T (kx x ky); rho_k (scalar-Y types)
  for (int k = 0; k < kx; k++) {
    S_ om[2] = {fzero<FrM<C>>(), fzero<FrM<C>>()};
    S_ rho = fzero<FrM<C>>();
    for (int j = 0; j < n; j++) {
      S_ psi = fzero<FrM<C>>();
      for (int i = 0; i < m; i++) psi = add(psi, mul(R[i * kx + k], G[i * n + j]));
      P[pm.PSI + k * n + j] = from_mont(psi);
      for (int l = 0; l < ky; l++) om[l] = add(om[l], mul(psi, S[j * ky + l]));
      if (ys) rho = add(rho, mul(psi, ys[j]));
    }
    for (int l = 0; l < ky; l++) P[pm.OM + k * ky + l] = from_mont(sub(om[l], T[l * kx + k]));
    if (bs) {
      for (int i = 0; i < m; i++) rho = add(rho, mul(R[i * kx + k], bs[i]));
      P[pm.RHO + k] = from_mont(rho);
    }
  }
  // Phi = S^T Gamma^T (ky x m); sigma_l (scalar-X types)
  for (int l = 0; l < ky; l++) {
    S_ sig = fzero<FrM<C>>();
    for (int i = 0; i < m; i++) {
      S_ phi = fzero<FrM<C>>();
      for (int j = 0; j < n; j++) phi = add(phi, mul(S[j * ky + l], G[i * n + j]));
      P[pm.PHI + l * m + i] = from_mont(phi);
      if (xs) sig = add(sig, mul(phi, xs[i]));
    }
    if (as) {
      for (int j = 0; j < n; j++) sig = add(sig, mul(S[j * ky + l], as[j]));
      P[pm.SIG + l] = from_mont(sig);
    }
  }
}
};

// The same for large arities (m n in the thousands and beyond, benches/bench.rs:451-498 uses 334 x 334): one lane per
// OUTPUT SCALAR instead of one per equation.  Phase a: conversions and the two matrix products (a lane = one entry
// of Psi or Phi = one inner product); phase b: the <= 8 scalars per equation that depend on Psi / Phi.
template <class C>
struct k_prep_prove_wide_a {
  static __device__ __forceinline__ void run(size_t g, size_t total, int W, int m, int n, int kx, int ky, const Fr<C>* G, const Fr<C>* R,
                        const Fr<C>* S, const Fr<C>* T, const Fr<C>* xs, const Fr<C>* ys, PoolMap pm, Fr<C>* pool,
                        int shared_vars) {
  if (g >= total) return;
  size_t e = g / W;
  int w = (int)(g % W);
  typedef Fr<C> S_;
  const size_t ev = shared_vars ? 0 : e;
  G += e * m * n;
  R += ev * m * kx;
  S += ev * n * ky;
  T += e * ky * kx;
  if (xs) xs += ev * m;
  if (ys) ys += ev * n;
  S_* P = pool + e * pm.total;
  int o = 0;
  if (w < o + m * kx) { P[pm.RC + w - o] = from_mont(R[w - o]); return; }
  o += m * kx;
  if (w < o + n * ky) { P[pm.SC + w - o] = from_mont(S[w - o]); return; }
  o += n * ky;
  if (w < o + ky * kx) { P[pm.TC + w - o] = from_mont(T[w - o]); return; }
  o += ky * kx;
  if (w < o + m) { if (xs) P[pm.XC + w - o] = from_mont(xs[w - o]); return; }
  o += m;
  if (w < o + n) { if (ys) P[pm.YC + w - o] = from_mont(ys[w - o]); return; }
  o += n;
  if (w < o + kx * n) {  // Psi[k][j] = sum_i R[i][k] G[i][j]
    int k = (w - o) / n, j = (w - o) % n;
    S_ psi = fzero<FrM<C>>();
    for (int i = 0; i < m; i++) psi = add(psi, mul(R[i * kx + k], G[i * n + j]));
    P[pm.PSI + k * n + j] = from_mont(psi);
    return;
  }
  o += kx * n;
  if (w < o + ky * m) {  // Phi[l][i] = sum_j S[j][l] G[i][j]
    int l = (w - o) / m, i = (w - o) % m;
    S_ phi = fzero<FrM<C>>();
    for (int j = 0; j < n; j++) phi = add(phi, mul(S[j * ky + l], G[i * n + j]));
    P[pm.PHI + l * m + i] = from_mont(phi);
  }
}
};
template <class C>
struct k_prep_prove_wide_b {
  static __device__ __forceinline__ void run(size_t g, size_t total, int m, int n, int kx, int ky, const Fr<C>* R, const Fr<C>* S, const Fr<C>* T,
                        const Fr<C>* xs, const Fr<C>* ys, const Fr<C>* as, const Fr<C>* bs, PoolMap pm, Fr<C>* pool,
                        int shared_vars) {
  if (g >= total) return;
  const int W = kx * ky + kx + ky;
  size_t e = g / W;
  int w = (int)(g % W);
  typedef Fr<C> S_;
  const size_t ev = shared_vars ? 0 : e;
  R += ev * m * kx;
  S += ev * n * ky;
  T += e * ky * kx;
  if (xs) xs += ev * m;
  if (ys) ys += ev * n;
  if (as) as += e * n;
  if (bs) bs += e * m;
  S_* P = pool + e * pm.total;
  if (w < kx * ky) {  // Omega[k][l] = sum_j Psi[k][j] S[j][l] - T[l][k]
    int k = w / ky, l = w % ky;
    S_ om = fzero<FrM<C>>();
    for (int j = 0; j < n; j++) om = add(om, mul(to_mont(P[pm.PSI + k * n + j]), S[j * ky + l]));
    P[pm.OM + k * ky + l] = from_mont(sub(om, T[l * kx + k]));
  } else if (w < kx * ky + kx) {  // rho_k = sum_j Psi[k][j] y_j + sum_i R[i][k] b_i   (scalar-Y types)
    int k = w - kx * ky;
    if (!bs) return;
    S_ rho = fzero<FrM<C>>();
    if (ys)
      for (int j = 0; j < n; j++) rho = add(rho, mul(to_mont(P[pm.PSI + k * n + j]), ys[j]));
    for (int i = 0; i < m; i++) rho = add(rho, mul(R[i * kx + k], bs[i]));
    P[pm.RHO + k] = from_mont(rho);
  } else {  // sigma_l = sum_i Phi[l][i] x_i + sum_j S[j][l] a_j   (scalar-X types)
    int l = w - kx * ky - kx;
    if (!as) return;
    S_ sig = fzero<FrM<C>>();
    if (xs)
      for (int i = 0; i < m; i++) sig = add(sig, mul(to_mont(P[pm.PHI + l * m + i]), xs[i]));
    for (int j = 0; j < n; j++) sig = add(sig, mul(S[j * ky + l], as[j]));
    P[pm.SIG + l] = from_mont(sig);
  }
}
};
// out[e * out_stride + i] = canonical(in[e * cnt + i]): the Gamma conversion of the verifier, one lane per scalar
template <class C>
struct k_fr_canonical {
  static __device__ __forceinline__ void run(size_t g, size_t total, int cnt, const Fr<C>* in, int out_stride, Fr<C>* out) {
  if (g >= total) return;
  size_t e = g / cnt;
  int i = (int)(g % cnt);
  out[e * out_stride + i] = from_mont(in[g]);
}
};

// canonical -> Montgomery boundary form (gs_fr_matmul hands the prover's canonical products back as Fr values)
template <class C>
__global__ void __launch_bounds__(64, GS_WPE) k_fr_to_mont(size_t total, const Fr<C>* in, Fr<C>* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  out[g] = to_mont(in[g]);
}

// Fr preparation for verify: Gamma (and scalar constants / target) -> canonical
template <class C>
struct k_prep_verify {
  static __device__ __forceinline__ void run(size_t e_in, size_t N, int m, int n, const Fr<C>* G, const Fr<C>* as,
                                                    const Fr<C>* bs, const Fr<C>* tq, PoolMap pm, Fr<C>* pool) {
  size_t e = e_in;
  if (e >= N) return;
  Fr<C>* P = pool + e * pm.total;
  if (G)  // (large arities convert Gamma with k_fr_canonical, one lane per scalar)
    for (int i = 0; i < m * n; i++) P[pm.GC + i] = from_mont(G[e * m * n + i]);
  if (as)
    for (int j = 0; j < n; j++) P[pm.AC + j] = from_mont(as[e * n + j]);
  if (bs)
    for (int i = 0; i < m; i++) P[pm.BC + i] = from_mont(bs[e * m + i]);
  if (tq) P[pm.NT] = from_mont(neg(tq[e]));
}
};

template <class C> GS_HD_NOINLINE void f12_pow_u64(Fp12<C>& r, const Fp12<C>& b, uint64_t k) {
  Fp12<C> acc;
  f12_one(acc);
  bool started = false;
  for (int i = 63; i >= 0; i--) {
    if (started) f12_sqr(acc, acc);
    if ((k >> i) & 1) {
      if (started)
        f12_mul(acc, acc, b);
      else
        acc = b;
      started = true;
    }
  }
  r = acc;
}

// Fr preparation for the batched verifier with the random exponents folded into the G1
// arguments: rho[e][2a+b] (64-bit) -> canonical RH[2a+b]; x-scalar types also need
// a_j*rho_ab (AR[(j*2+a)*2+b]); QuadEqu needs (-t)*rho_ab (NR[a*2+b]).
template <class C>
__global__ void __launch_bounds__(64, GS_WPE)
    k_prep_verify_rlc(size_t N, int m, int n, const Fr<C>* G, const Fr<C>* as, const Fr<C>* bs, const Fr<C>* tq,
                      const uint64_t* rho, PoolMap pm, Fr<C>* pool) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  Fr<C>* P = pool + e * pm.total;
  if (G)
    for (int i = 0; i < m * n; i++) P[pm.GC + i] = from_mont(G[e * m * n + i]);
  if (bs)
    for (int i = 0; i < m; i++) P[pm.BC + i] = from_mont(bs[e * m + i]);
  Fr<C> rm[4];
  for (int c = 0; c < 4; c++) {
    Fr<C> r = fzero<FrM<C>>();
    uint64_t v = rho[e * 4 + c];
    r.v[0] = (uint32_t)v;
    r.v[1] = (uint32_t)(v >> 32);
    P[pm.RH + c] = r;      // canonical (rho < 2^64 < r)
    rm[c] = to_mont(r);
  }
  if (as)
    for (int j = 0; j < n; j++)
      for (int c = 0; c < 4; c++) P[pm.AR + j * 4 + c] = from_mont(mul(as[e * n + j], rm[c]));
  if (tq) {
    Fr<C> nt = neg(tq[e]);
    for (int c = 0; c < 4; c++) P[pm.NR + c] = from_mont(mul(nt, rm[c]));
  }
}

// out_t[e] = t_e^(rho[e][3]) for the PPE target cell (boundary in, internal out)
template <class C>
__global__ void __launch_bounds__(64, GS_WPE) k_rlc_tpow(size_t N, const uint8_t* target, const uint64_t* rho,
                                                 Fp12<C>* out_t) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  Fp12<C> t, h;
  f12_from_boundary<C>(t, reinterpret_cast<const BFq<C>*>(target) + 12 * e);
  f12_pow_u64(h, t, rho[e * 4 + 3]);
  out_t[e] = h;
}

// --------------------------------------------------------------------------
// linear-combination engine
// --------------------------------------------------------------------------
template <class C, class F, bool ENDO = true>
struct k_var {
  static __device__ __forceinline__ void run(size_t g, size_t total, int ntask, const VarTask* tasks, ArrTab arrs,
                                            const Fr<C>* pool, int pool_n, Jac<F>* part, int nslots) {
  if (g >= total) return;
  size_t e = g / ntask;
  VarTask t = tasks[g % ntask];
  Fr<C> k = pool[e * pool_n + t.s_idx];
  Aff<F> P;
  aff_load<C>(P, arrs.base[t.p_arr] + e * arrs.stride[t.p_arr] + (size_t)t.p_idx * AFFB(C, F));
  if (t.neg) P.y = neg(P.y);
  Jac<F> J;
  jac_smul_any<C, F, ENDO>(J, P, k);
  part[e * nslots + t.slot] = J;
}
};

// joint MSM: one lane = one GrpTask (<= TMAX bases sharing a doubling chain; the lane's table build serves all of the
// group's outputs, which run one after the other)
// WPE2 = 2 (G1 only): the kernel is built for TWO waves per SIMD.  With the point operations as register-only
// subroutines (gs_pointops_asm.h: v48 .. v206 on BLS12-381) the whole lane fits 256 registers, and a second resident
// wave issues into every slot the first one leaves empty -- at one wave per SIMD a v_mad_u64_u32 issues every ~5.9
// cycles and a full-rate instruction every ~5.3, at two every ~4.75 / ~2.7 (tools/ubench.hip, profiles/r4/).  Planned
// when the launch has enough waves to put two on every SIMD (csrc/gs_amd.hip, var_w2).
template <class C, class F, int TMAX, int W, int WPE2 = 1>
struct k_var_multi {
  static constexpr int WPE = WPE2;
  static __device__ __forceinline__ void run(size_t w_in, size_t total, int ngrp, const GrpTask* grps, const VarTask* tasks,
                                                  ArrTab arrs, const Fr<C>* pool, int pool_n, Jac<F>* part,
                                                  int nslots, size_t g0, Aff<F>* tabws) {
  size_t w = w_in;  // lane within this launch = its slot in `tabws`
  size_t g = g0 + w;
  if (g >= total) return;
  size_t e = g / ngrp;
  GrpTask gt = grps[g % ngrp];
  // the lane's workspace: its affine table, then the Jacobian staging of the build (STRAUS_WS_BYTES per lane)
  constexpr size_t NTAB = (size_t)TMAX << (W - 1);
  uint8_t* ws = reinterpret_cast<uint8_t*>(tabws) + w * (NTAB * (sizeof(Aff<F>) + sizeof(Jac<F>)));
  Aff<F>* at = reinterpret_cast<Aff<F>*>(ws);
  Jac<F>* jt = reinterpret_cast<Jac<F>*>(ws + NTAB * sizeof(Aff<F>));
  F zback;
  {
    Aff<F> P[TMAX];
    for (uint32_t i = 0; i < gt.nt; i++) {
      VarTask t = tasks[gt.first[0] + i];
      aff_load<C>(P[i], arrs.base[t.p_arr] + e * arrs.stride[t.p_arr] + (size_t)t.p_idx * AFFB(C, F));
      if (t.neg) P[i].y = neg(P[i].y);
    }
    unsigned long long sb0 = GS_STAMP_T();
    jac_straus_build<C, F, TMAX, W>(at, zback, P, (int)gt.nt, jt);
    GS_STAMP_ADD(5, GS_STAMP_T() - sb0);
  }
  for (uint32_t o = 0; o < gt.no; o++) {
    Fr<C> k[TMAX];
    for (uint32_t i = 0; i < gt.nt; i++) k[i] = pool[e * pool_n + tasks[gt.first[o] + i].s_idx];
    Jac<F> J;
    jac_straus_run<C, F, TMAX, W, true, (WPE2 == 2 ? 18 : 36)>(J, k, (int)gt.nt, at, zback);
    part[e * nslots + gt.slot[o]] = J;
  }
}
};

// ---- large arities: window tables of the BASES, shared by every output that uses them --------------------------------
// The verifier's Gamma^T c is 2 n outputs over the SAME 2 m commitment components (benches/bench.rs:451-498: m = n = 334).
// A Straus lane builds its own tables (8 bases x 16 entries, shared by at most 4 outputs); with hundreds of outputs per
// base it pays to build ONE wide table per (equation, base) -- 2^(W-1) = 128 true-affine multiples for 8-bit signed
// windows, one inversion per base -- and let every output's lanes read them: 2 x 17 look-ups and mixed additions per
// term instead of 2 x 27, one shared doubling chain of 128 instead of 130 doublings, no per-lane build at all.
template <class C, class F, int W>
struct k_tab_build {
  static __device__ __forceinline__ void run(size_t g, size_t total, int nb, ArrTab arrs, int arr, Jac<F>* stage, Aff<F>* tabs) {
  if (g >= total) return;
  constexpr int NE = 1 << (W - 1);
  size_t e = g / nb;
  int b = (int)(g % nb);
  Aff<F> P;
  aff_load<C>(P, arrs.base[arr] + e * arrs.stride[arr] + (size_t)b * AFFB(C, F));
  smul_affine_table(tabs + g * NE, stage + g * NE, P, NE);
}
};
// one lane = one GrpTask as in k_var_multi (<= TMAX terms sharing the doubling chain, `no` outputs one after the other);
// base t of the group is entry tasks[..].p_idx of array 0, its table tabs[(e * nb + p_idx) * NE ..]
template <class C, class F, int TMAX, int W>
struct k_var_tab {
  static __device__ __forceinline__ void run(size_t g, size_t total, int ngrp, const GrpTask* grps, const VarTask* tasks,
                                              const Fr<C>* pool, int pool_n, Jac<F>* part, int nslots, const Aff<F>* tabs,
                                              int nb) {
  if (g >= total) return;
  constexpr int NE = 1 << (W - 1);
  size_t e = g / ngrp;
  GrpTask gt = grps[g % ngrp];
  for (uint32_t o = 0; o < gt.no; o++) {
    Fr<C> k[TMAX];
    const Aff<F>* tp[TMAX];
    uint32_t negm = 0;
    for (uint32_t i = 0; i < gt.nt; i++) {
      VarTask t = tasks[gt.first[o] + i];
      k[i] = pool[e * pool_n + t.s_idx];
      tp[i] = tabs + (e * (size_t)nb + t.p_idx) * NE;
      if (t.neg) negm |= 1u << i;
    }
    Jac<F> J;
    jac_straus_run<C, F, TMAX, W, true>(J, k, (int)gt.nt, (const Aff<F>*)nullptr, one_of<F>(), tp, negm);
    part[e * nslots + gt.slot[o]] = J;
  }
}
};

template <class C, class F>
struct k_fix {
  static __device__ __forceinline__ void run(size_t g, size_t total, int ntask, const FixTask* tasks, ArrTab arrs,
                                            const Fr<C>* pool, int pool_n, const Aff<F>* tab, Jac<F>* part,
                                            int nslots) {
  if (g >= total) return;
  size_t e = g / ntask;
  FixTask t = tasks[g % ntask];
  Jac<F> acc;
  jac_set_inf(acc);
  // 2 x 16 window look-ups (16-bit windows, k_build_tables16) and the optional affine addend = up to 33 mixed
  // additions.  The tables are 0.6 / 1.2 GB per CRS: every look-up is an HBM access, and at one wave per SIMD its whole
  // latency is exposed unless it was requested a step ahead -- so the entry of step j + 1 is loaded (global_load:
  // vmcnt only) before the addition of step j starts; the running sum is updated in place (G1: the register-only
  // subroutines of gs_pointops_asm.h) and never has its address taken.
  typedef const __attribute__((address_space(1))) limb_t* gptr;
  auto ldaff = [](const Aff<F>* p) -> Aff<F> {
    Aff<F> v;
    constexpr int NWD = (int)(sizeof(Aff<F>) / sizeof(limb_t));
    gptr w = (gptr) reinterpret_cast<const limb_t*>(p);
#pragma unroll
    for (int q = 0; q < NWD; q++) reinterpret_cast<limb_t*>(&v)[q] = w[q];
    return v;
  };
  Fr<C> k0, k1;  // the (<= 2) scalars of this lane; an absent term keeps zero digits
  for (int i = 0; i < FrM<C>::N; i++) k0.v[i] = k1.v[i] = 0;
  if (t.t0 != 0xFF) k0 = pool[e * pool_n + t.s0];
  if (t.t1 != 0xFF) k1 = pool[e * pool_n + t.s1];
  auto slot = [&](int j) -> const Aff<F>* {  // j = term * 16 + window; nullptr = nothing to add
    const int term = j >> 4, w = j & 15;
    const Fr<C>& k = term ? k1 : k0;
    const uint32_t d = (k.v[w >> 1] >> ((w & 1) * 16)) & 65535u;
    const int tb = term ? t.t1 : t.t0;
    if (d == 0 || tb == 0xFF) return nullptr;
    return tab + ((size_t)tb * 16 + (size_t)w) * 65536 + d;
  };
  const Aff<F>* cur = slot(0);
  Aff<F> q = ldaff(cur ? cur : tab);
#pragma unroll 1
  for (int j = 0; j < 32; j++) {
    const Aff<F>* nxt = j + 1 < 32 ? slot(j + 1) : nullptr;
    Aff<F> qn = ldaff(nxt ? nxt : tab);  // (a step without a look-up requests entry 0: harmless, never used)
    if (cur) jac_madd_ip(acc, q);
    q = qn;
    cur = nxt;
  }
  if (t.a_arr != 0xFF) {
    Aff<F> a;
    aff_load<C>(a, arrs.base[t.a_arr] + e * arrs.stride[t.a_arr] + (size_t)t.a_idx * AFFB(C, F));
    if (t.a_neg) a.y = neg(a.y);
    jac_madd_ip(acc, a);
  }
  part[e * nslots + t.slot] = acc;
}
};

// One lane = up to RED_K consecutive outputs (two points each) and ONE inversion for all of them (Montgomery's trick
// over their Z coordinates): the inversion is most of this kernel (496 of ~530 Fq multiplications per output in G1).
constexpr int RED_K = 8;
template <class C, class F>
struct k_red {
  static __device__ __forceinline__ void run(size_t g, size_t total, int ntask, const RedTask* tasks, const Jac<F>* part,
                                            int nslots, OutTab outs, int K) {
  if (g >= total) return;
  const int nl = (ntask + K - 1) / K;  // lanes per equation
  size_t e = g / nl;
  int t0 = (int)(g % nl) * K, nt = ntask - t0 < K ? ntask - t0 : K;
  const Jac<F>* P = part + e * nslots;
  Jac<F> s[2 * RED_K];
  F pre[2 * RED_K];  // pre[i] = z_0 ... z_(i-1) (identities count as 1)
  F acc = one_of<F>();
  for (int k = 0; k < nt; k++) {
    RedTask t = tasks[t0 + k];
    Jac<F> s0 = P[t.b0], s1 = P[t.b1];
    for (uint32_t i = t.b0 + 1; i < t.e0; i++) jac_add(s0, s0, P[i]);
    for (uint32_t i = t.b1 + 1; i < t.e1; i++) jac_add(s1, s1, P[i]);
    s[2 * k] = s0;
    s[2 * k + 1] = s1;
    pre[2 * k] = acc;
    if (!is_zero_limbs(s0.z)) acc = mul(acc, s0.z);
    pre[2 * k + 1] = acc;
    if (!is_zero_limbs(s1.z)) acc = mul(acc, s1.z);
  }
  F suf = inv(acc);  // 1 / (z_0 ... z_(2 nt - 1))
  for (int i = 2 * nt - 1; i >= 0; i--) {
    RedTask t = tasks[t0 + i / 2];
    Aff<F> a;
    F zi = mul(suf, pre[i]);  // 1 / z_i
    if (!is_zero_limbs(s[i].z)) suf = mul(suf, s[i].z);
    jac_to_aff_zinv(a, s[i], zi);  // (identity -> (0, 0), zi unused)
    uint8_t* o = outs.base[t.out_arr] + e * outs.stride[t.out_arr] + (2 * (size_t)t.out_idx + (i & 1)) * AFFB(C, F);
    aff_store<C>(o, a);
  }
}
};

// Large arities (benches/bench.rs:451-498, m = n = 334): an output is the sum of hundreds of partial slots.  k_red
// folds its slots serially in ONE lane, so the host first folds runs of K slots in parallel (one lane per run), and
// again, until every run k_red sees is short: a segmented K-ary tree reduction.
struct FoldTask {
  uint32_t lo, hi, dst, pad;  // dst slot (in the output array) <- sum of input slots [lo, hi)
};
template <class C, class F>
struct k_slot_fold {
  static __device__ __forceinline__ void run(size_t g, size_t total, int ntask, const FoldTask* tasks, const Jac<F>* in,
                                                          int ns_in, Jac<F>* out, int ns_out) {
  if (g >= total) return;
  size_t e = g / ntask;
  FoldTask t = tasks[g % ntask];
  const Jac<F>* P = in + e * ns_in;
  Jac<F> s = P[t.lo];
  for (uint32_t i = t.lo + 1; i < t.hi; i++) jac_add(s, s, P[i]);
  out[e * ns_out + t.dst] = s;
}
};

// --------------------------------------------------------------------------
// pairing side
// --------------------------------------------------------------------------
// Line tables of the six CRS G2 points (v0.0 v0.1 v1.0 v1.1 W2.0 W2.1), built once per CRS: one lane per point.
template <class C> __global__ void __launch_bounds__(64, GS_WPE) k_line_tables(const uint8_t* crs_g2, Line<C>* out) {
  int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= 6) return;
  Aff<Fp2<C>> q;
  aff_load<C>(q, crs_g2 + (size_t)i * AFFB(C, Fp2<C>));
  miller_line_table<C>(out + (size_t)i * miller_line_count<C>(), q);
}

// Lanes are TASK-MAJOR: lane g works on task g / N of equation g % N, so a wave holds 64 equations of one task and its
// lanes agree on the number of pairs and on which of them read line tables (`ltab`, pairs with Q array 2 = CRS).
template <class C, bool TWIN>
struct k_miller {
  static __device__ __forceinline__ void run(size_t g, size_t total, int ntask, const MillerTask* tasks, ArrTab parr,
                                               ArrTab qarr, Fp12<C>* out, int ostride, const Line<C>* ltab) {
  if (g >= total) return;
  const size_t N = total / ntask;
  const size_t ti = g / N, e = g % N;
  const size_t go = e * ntask + ti;  // partials stay equation-major for k_final
  MillerTask t = tasks[ti];
  Aff<Fq<C>> p0[MILLER_CH];
  Aff<Fp2<C>> qs[MILLER_CH];
  Proj2<C> ts[MILLER_CH];
  const Line<C>* fx[MILLER_CH];
  bool anyfx = false;  // uniform over the wave (task-major lanes)
  for (int k = 0; k < t.np; k++) {
    fx[k] = (ltab && t.pr[k].q_arr == 2) ? ltab + (size_t)t.pr[k].q_idx * miller_line_count<C>() : nullptr;
    anyfx |= fx[k] != nullptr;
  }
  if constexpr (TWIN) {
    Aff<Fq<C>> p1[MILLER_CH];
    uint8_t live[MILLER_CH];
    for (int k = 0; k < t.np; k++) {
      PairRef r = t.pr[k];
      const uint8_t* pb = parr.base[r.p_arr] + e * parr.stride[r.p_arr] + (size_t)r.p_idx * AFFB(C, Fq<C>);
      aff_load<C>(p0[k], pb);
      aff_load<C>(p1[k], pb + AFFB(C, Fq<C>));
      aff_load<C>(qs[k], qarr.base[r.q_arr] + e * qarr.stride[r.q_arr] + (size_t)r.q_idx * AFFB(C, Fp2<C>));
      if (r.neg) {
        p0[k].y = neg(p0[k].y);
        p1[k].y = neg(p1[k].y);
      }
    }
    Fp12<C> f0, f1;
    multi_miller2(f0, f1, p0, p1, qs, t.np, ts, live, anyfx ? fx : nullptr);
    out[2 * go] = f0;      // cell (0, b)
    out[2 * go + 1] = f1;  // cell (1, b)
  } else {
    bool live[MILLER_CH];
    for (int k = 0; k < t.np; k++) {
      PairRef r = t.pr[k];
      aff_load<C>(p0[k], parr.base[r.p_arr] + e * parr.stride[r.p_arr] + (size_t)r.p_idx * AFFB(C, Fq<C>));
      aff_load<C>(qs[k], qarr.base[r.q_arr] + e * qarr.stride[r.q_arr] + (size_t)r.q_idx * AFFB(C, Fp2<C>));
      if (r.neg) p0[k].y = neg(p0[k].y);
    }
    Fp12<C> f;
    multi_miller(f, p0, qs, t.np, ts, live, anyfx ? fx : nullptr);
    out[(size_t)ostride * go] = f;  // the task's own cell (slot 0 of 2), or densely packed (batched verifier)
  }
}
};

// ---- pair-cooperative twin (multi_miller_pair): two lanes per (equation, task), one accumulator each ----------------
// Exchange policies: how a lane hands its tangent / chord line (6 L dwords) to its partner lane ^ 1 (put / get, see
// multi_miller_pair).
//  * PairLds: a 16-byte-interleaved LDS slot per lane, [6 L / 4][64] x int4 (conflict-free ds_write_b128 / ds_read_b128,
//    21 + 21 instructions for BLS12-381); the block is ONE wave, so program order is LDS order and the barriers below
//    are compiler fences; the partner's line is fetched only after the lane's own line product;
//  * PairDpp: one v_mov_b32 with quad_perm [1,0,3,2] per dword at put(), no memory at all.
// Measured at 2^16 (one box, alternating).  First shape of the round loop: DPP 157.1-158.9 ms, LDS 161.3-161.6 ms.
// Final shape (the products of a round in one lambda, gs_pairing.cuh): LDS **147.5-147.7 ms**, DPP 154.6-156.0 ms on
// BLS12-381 (a line parked in LDS frees 84 registers across the lane's own product); BN254 100.5 (DPP) against
// 101.3 ms (LDS).  The planner takes LDS on BLS12-381 and DPP on BN254 (profiles/r3/ab_exchange.txt).
template <class C> GS_HD int32_t& line_word(Line<C>& l, int i) {
  constexpr int L = C::L;
  const int c = i / L, j = i % L;
  return c == 0 ? l.l0.c0.v[j] : c == 1 ? l.l0.c1.v[j] : c == 2 ? l.lx.c0.v[j] : c == 3 ? l.lx.c1.v[j]
         : c == 4 ? l.ly.c0.v[j] : l.ly.c1.v[j];
}
template <class C> struct PairLds {
  int4* slots;  // [6 L / 4][64]
  int lane;
  static constexpr int W = 6 * C::L / 4;
  static_assert(6 * C::L % 4 == 0, "a line is a whole number of 16-byte words");
  __device__ __forceinline__ void put(const Line<C>& mine) const {
    Line<C> m = mine;
    __syncthreads();  // (the partner has read the previous line: one wave per block, this only orders the compiler)
#pragma unroll
    for (int q = 0; q < W; q++)
      slots[q * 64 + lane] = make_int4(line_word(m, 4 * q), line_word(m, 4 * q + 1), line_word(m, 4 * q + 2),
                                       line_word(m, 4 * q + 3));
    __syncthreads();
  }
  __device__ __forceinline__ Line<C> get() const {
    Line<C> r;
#pragma unroll
    for (int q = 0; q < W; q++) {
      int4 v = slots[q * 64 + (lane ^ 1)];
      line_word(r, 4 * q) = v.x;
      line_word(r, 4 * q + 1) = v.y;
      line_word(r, 4 * q + 2) = v.z;
      line_word(r, 4 * q + 3) = v.w;
    }
    return r;
  }
};
template <class C> struct PairDpp {
  Line<C> got;  // the exchange happens at put(): both lines are in registers from then on
  __device__ __forceinline__ void put(const Line<C>& mine) {
    Line<C> m = mine;
#pragma unroll
    for (int i = 0; i < 6 * C::L; i++)
      line_word(got, i) = __builtin_amdgcn_mov_dpp(line_word(m, i), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
  }
  __device__ __forceinline__ Line<C> get() const { return got; }
};
// Lanes are pair-major inside task-major: lane g works on component g & 1 of pair g >> 1 = (task, equation); a wave is
// 32 equations of one task.  Same task tables and the same output layout as the twin kernel (out[2 go + a]); the
// stepping triples of a task come first in its list (chunk_tasks), the table-reading ones after them.
template <class C, bool DPP>
struct k_miller_pair {
  static __device__ __forceinline__ void run(size_t g, size_t total, int ntask, const MillerTask* tasks, ArrTab parr,
                                                    ArrTab qarr, Fp12<C>* out, const Line<C>* ltab) {
  __shared__ int4 xslots[DPP ? 1 : (6 * C::L / 4) * 64];
  if (g >= total) return;  // total is even: a pair leaves together
  const int a = (int)(g & 1);
  const size_t gp = g >> 1, N = total / (2 * (size_t)ntask);
  const size_t ti = gp / N, e = gp % N;
  const size_t go = e * ntask + ti;
  MillerTask t = tasks[ti];
  constexpr int OWN = (MILLER_CH + 1) / 2;
  Aff<Fq<C>> ps[MILLER_CH];
  Aff<Fp2<C>> qown[OWN];
  Proj2<C> ts[OWN];
  const Line<C>* fx[MILLER_CH];
  int nstep = 0;
  for (int k = 0; k < t.np; k++) {
    fx[k] = (ltab && t.pr[k].q_arr == 2) ? ltab + (size_t)t.pr[k].q_idx * miller_line_count<C>() : nullptr;
    if (!fx[k]) nstep = k + 1;  // (a prefix by construction)
  }
  uint32_t qok = 0;
  for (int k = 0; k < t.np; k++) {
    PairRef r = t.pr[k];
    aff_load<C>(ps[k], parr.base[r.p_arr] + e * parr.stride[r.p_arr] + ((size_t)r.p_idx + a) * AFFB(C, Fq<C>));
    if (r.neg) ps[k].y = neg(ps[k].y);
    // identity test of Q on the boundary words ((0, 0) is the flag, and the Montgomery form of 0 is 0)
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(qarr.base[r.q_arr] + e * qarr.stride[r.q_arr] +
                                                           (size_t)r.q_idx * AFFB(C, Fp2<C>));
    uint32_t any = 0;
    for (int w = 0; w < (int)(AFFB(C, Fp2<C>) / 4); w++) any |= qw[w];
    if (any) qok |= 1u << k;
  }
  const int rounds = (nstep + 1) / 2;
  for (int r = 0; r < rounds; r++) {
    int k = 2 * r + a;
    if (k >= nstep) k = nstep - 1;  // odd count: lane 1 steps a copy of the last point, never consumed
    PairRef pr = t.pr[k];
    aff_load<C>(qown[r], qarr.base[pr.q_arr] + e * qarr.stride[pr.q_arr] + (size_t)pr.q_idx * AFFB(C, Fp2<C>));
  }
  Fp12<C> f;
  if constexpr (DPP) {
    PairDpp<C> x;
    multi_miller_pair(f, a, ps, qown, qok, nstep, (int)t.np, ts, fx, x);
  } else {
    PairLds<C> x{xslots, (int)threadIdx.x};
    multi_miller_pair(f, a, ps, qown, qok, nstep, (int)t.np, ts, fx, x);
  }
  out[2 * go + a] = f;
}
};

// Cell c = 2a + b: product of its Miller partials (CellMap), final exponentiation,
// compare with 1 or the PPE target (verifier.rs:50-53) -> cellok[e*4+c].
template <class C> GS_HD_NOINLINE void cell_product(Fp12<C>& f, const Fp12<C>* mpart, size_t e, int ntask, int c,
                                                    const CellMap& cm) {
  int lo = cm.lo[c], hi = cm.hi[c], a = cm.sub[c];
  f = mpart[2 * (e * ntask + lo) + a];
  for (int i = lo + 1; i < hi; i++) f12_mul(f, f, mpart[2 * (e * ntask + i) + a]);
}
// Large arities: a cell has hundreds of Miller partials.  One lane per (equation, cell, run of K partials) multiplies
// its run; repeated by the host until k_final's own serial product is short (segmented K-ary tree in GT).
// Output layout = cell_product's: out[2 * (e * nt_out + cm_out.lo[c] + run) + 0].
template <class C>
struct k_cell_fold {
  static __device__ __forceinline__ void run(size_t g, size_t total, int runs_max, int ntask_in, CellMap cm_in,
                                                          const Fp12<C>* in, int K, int nt_out, CellMap cm_out,
                                                          Fp12<C>* out) {
  if (g >= total) return;
  int run = (int)(g % runs_max);
  int c = (int)((g / runs_max) & 3);
  size_t e = g / ((size_t)runs_max * 4);
  int lo = cm_in.lo[c] + run * K, hi = lo + K < cm_in.hi[c] ? lo + K : cm_in.hi[c], a = cm_in.sub[c];
  if (lo >= cm_in.hi[c]) return;
  Fp12<C> f = in[2 * (e * ntask_in + lo) + a];
  for (int i = lo + 1; i < hi; i++) f12_mul(f, f, in[2 * (e * ntask_in + i) + a]);
  out[2 * (e * nt_out + cm_out.lo[c] + run)] = f;
}
};

template <class C>
struct k_final {
  static __device__ __forceinline__ void run(size_t g, size_t N, int ntask, CellMap cm, const Fp12<C>* mpart,
                                              const uint8_t* target, uint8_t* cellok) {
  if (g >= N * 4) return;
  size_t e = g >> 2;
  int c = (int)(g & 3);
  Fp12<C> f;
  unsigned long long sf0 = GS_STAMP_T();
  cell_product(f, mpart, e, ntask, c, cm);
  unsigned long long sf1 = GS_STAMP_T();
  Fp12<C> r;
  final_exp(r, f);
  unsigned long long sf2 = GS_STAMP_T();
  bool ok;
  if (c == 3 && target) {
    Fp12<C> t;
    f12_from_boundary<C>(t, reinterpret_cast<const BFq<C>*>(target) + 12 * e);
    ok = f12_eq(r, t);
  } else {
    ok = f12_is_one(r);
  }
  cellok[g] = ok ? 1 : 0;
  GS_STAMP_ADD(6, sf1 - sf0);
  GS_STAMP_ADD(7, sf2 - sf1);
  GS_STAMP_ADD(8, GS_STAMP_T() - sf2);
  GS_STAMP_ADD(9, 1);
}
};

// The same with a 3-lane group per cell (gs_coop.cuh) for batches that cannot fill the chip with one lane per
// final exponentiation.  blockDim.x == 63: one wave = 21 groups; idle groups of the last wave shadow the last cell so
// that every lane reaches the shuffles.
template <class C>
struct k_final_coop {
  static __device__ __forceinline__ void run(size_t gt, size_t N, int ntask, CellMap cm, const Fp12<C>* mpart,
                                                   const uint8_t* target, uint8_t* cellok) {
  int lane = (int)(gt % 63), j = lane % 3;  // 63-thread blocks: gt % 63 is the thread within its block (= wave)
  size_t g = (gt / 63) * 21 + lane / 3;
  bool live = g < N * 4;
  if (!live) g = N * 4 - 1;
  size_t e = g >> 2;
  int c = (int)(g & 3);
  Fp12<C> f;
  cell_product(f, mpart, e, ntask, c, cm);
  Fp12<C> r;
  CoopWave xw{lane - j};
  ExpXCoop<C, CoopWave> ex{j, &xw};
  final_exp_with(r, f, ex);
  bool ok;
  if (c == 3 && target) {
    Fp12<C> t;
    f12_from_boundary<C>(t, reinterpret_cast<const BFq<C>*>(target) + 12 * e);
    ok = f12_eq(r, t);
  } else {
    ok = f12_is_one(r);
  }
  if (live && j == 0) cellok[g] = ok ? 1 : 0;
}
};

struct k_and4 {
  static __device__ __forceinline__ void run(size_t e_in, size_t N, const uint8_t* cellok, uint8_t* ok) {
  size_t e = e_in;
  if (e >= N) return;
  const uint8_t* c = cellok + e * 4;
  ok[e] = (c[0] & c[1] & c[2] & c[3]) ? 1 : 0;
}
};

// E::multi_pairing per row: k pairs -> Miller product -> final exponentiation
template <class C>
__global__ void __launch_bounds__(64, GS_WPE) k_multi_pairing(size_t n, int k, const uint8_t* P, const uint8_t* Q,
                                                      uint8_t* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  Aff<Fq<C>> ps[MILLER_CH];
  Aff<Fp2<C>> qs[MILLER_CH];
  Proj2<C> ts[MILLER_CH];
  bool live[MILLER_CH];
  Fp12<C> acc, f;
  f12_one(acc);
  for (int b = 0; b < k; b += MILLER_CH) {
    int np = k - b < MILLER_CH ? k - b : MILLER_CH;
    for (int i = 0; i < np; i++) {
      aff_load<C>(ps[i], P + (g * k + b + i) * AFFB(C, Fq<C>));
      aff_load<C>(qs[i], Q + (g * k + b + i) * AFFB(C, Fp2<C>));
    }
    multi_miller(f, ps, qs, np, ts, live);
    f12_mul(acc, acc, f);
  }
  final_exp(f, acc);
  f12_to_boundary<C>(reinterpret_cast<BFq<C>*>(out) + 12 * g, f);
}

// out[i] = base^(k[i]) for base in GT (cyclotomic squarings are valid)
template <class C>
__global__ void __launch_bounds__(64, GS_WPE) k_gt_pow(size_t n, const uint8_t* base, const Fr<C>* k, uint8_t* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  Fr<C> s = from_mont(k[g]);
  Fp12<C> b, acc;
  f12_from_boundary<C>(b, reinterpret_cast<const BFq<C>*>(base));
  f12_one(acc);
  bool started = false;
  int since = 0;
  for (int i = FrM<C>::BITS - 1; i >= 0; i--) {
    if (started) {
      f12_cyclo_sqr(acc, acc);
      since++;
    }
    if (get_bit(s, i)) {
      if (started)
        f12_mul(acc, acc, b);
      else
        acc = b;
      started = true;
      since = 0;
    } else if (since == 3) {
      f12_vreduce(acc);
      since = 0;
    }
  }
  f12_to_boundary<C>(reinterpret_cast<BFq<C>*>(out) + 12 * g, acc);
}

// ---- batched (random-linear-combination) verifier: tree product + one final exponentiation ----
// out[i] = product of in[i*K .. min((i+1)*K, n_in))
template <class C>
__global__ void __launch_bounds__(64, GS_WPE) k_gt_prod(size_t n_in, const Fp12<C>* in, size_t n_out, Fp12<C>* out,
                                                        int K) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_out) return;
  size_t lo = g * (size_t)K, hi = lo + K < n_in ? lo + K : n_in;
  Fp12<C> acc = in[lo];
  for (size_t i = lo + 1; i < hi; i++) f12_mul(acc, acc, in[i]);
  out[g] = acc;
}
// internal <-> boundary copies of GT arrays (accumulators cross the API in boundary form)
template <class C> __global__ void __launch_bounds__(64, GS_WPE) k_gt_export(size_t n, const Fp12<C>* in, uint8_t* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  f12_to_boundary<C>(reinterpret_cast<BFq<C>*>(out) + 12 * g, in[g]);
}
template <class C> __global__ void __launch_bounds__(64, GS_WPE) k_gt_import(size_t n, const uint8_t* in, Fp12<C>* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  Fp12<C> t;
  f12_from_boundary<C>(t, reinterpret_cast<const BFq<C>*>(in) + 12 * g);
  out[g] = t;
}
// acc[0] = Miller-side accumulator, acc[1] = target-side accumulator: ok = (FE(acc[0]) == acc[1])
// launched with ONE block of 3 lanes: a single 3-lane cooperative final exponentiation
template <class C> __global__ void k_fe_eq(const Fp12<C>* acc, uint8_t* ok) {
  int j = (int)threadIdx.x;
  Fp12<C> r;
  CoopWave xw{0};
  ExpXCoop<C, CoopWave> ex{j, &xw};
  final_exp_with(r, acc[0], ex);
  bool same = f12_eq(r, acc[1]);
  if (j == 0) ok[0] = same ? 1 : 0;
}
template <class C> __global__ void k_gt_set_one(Fp12<C>* p) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Fp12<C> o;
  f12_one(o);
  p[0] = o;
}

// --------------------------------------------------------------------------
// wire format (gs_wire.cuh): arrays of elements, one lane per element
// --------------------------------------------------------------------------
template <class C, class F> constexpr size_t wire_point_bytes(bool compressed) {
  return (size_t)(compressed ? 1 : 2) * NCoord<F>::V * C::N * 4;
}
template <class C, class F>
__global__ void __launch_bounds__(64, GS_WPE) k_wire_enc_pts(size_t n, const uint8_t* pts, int compressed, uint8_t* out) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  Aff<F> p;
  aff_load<C>(p, pts + g * AFFB(C, F));
  wire_encode_point<C, F>(out + g * wire_point_bytes<C, F>(compressed != 0), p, compressed != 0);
}
template <class C, class F>
__global__ void __launch_bounds__(64, GS_WPE) k_wire_dec_pts(size_t n, const uint8_t* in, int compressed, int validate,
                                                     uint8_t* pts, uint8_t* ok) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  Aff<F> p;
  bool good = wire_decode_point<C, F>(p, in + g * wire_point_bytes<C, F>(compressed != 0), compressed != 0, validate != 0);
  aff_store<C>(pts + g * AFFB(C, F), p);
  ok[g] = good ? 1 : 0;
}
// In-memory points (boundary limbs, include/gs_amd.h) through the checks the wire decoder applies to decoded ones:
// canonical coordinates (every Montgomery word string < p), the identity flag (0, 0), on the curve, and in the r-torsion
// (the endomorphism tests of gs_wire.cuh).  The reference's plain double-and-add takes ANY curve point
// (data_structures.rs:336-342); the GLV / psi-GLS scalar multiplications here are only defined on the r-torsion, so a
// caller that did not get its points from a validating decoder asks this first (or turns the endomorphisms off).
template <class C, class F>
__global__ void __launch_bounds__(64, GS_WPE) k_validate_pts(size_t n, const uint8_t* pts, uint8_t* ok) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  constexpr int NW = (int)(AFFB(C, F) / (4 * C::N));  // Fq words strings per point: 2 (G1) or 4 (G2)
  const uint32_t* w = reinterpret_cast<const uint32_t*>(pts + g * AFFB(C, F));
  bool canon = true, zero = true;
  for (int k = 0; k < NW; k++) {
    uint32_t t[C::N];
    for (int i = 0; i < C::N; i++) t[i] = w[k * C::N + i];
    canon = canon && words_lt_p<C>(t);
    zero = zero && words_zero<C::N>(t);
  }
  bool good = canon;
  if (canon && !zero) {
    Aff<F> p;
    aff_load<C>(p, pts + g * AFFB(C, F));
    good = eq(sqr(p.y), curve_rhs<C>(p.x)) && in_prime_subgroup<C>(p);
  }
  ok[g] = good ? 1 : 0;
}
// Fq arrays (GT = 12 per element): dir 0 boundary -> canonical little-endian bytes, dir 1 back (ok = canonical)
template <class C>
__global__ void __launch_bounds__(64, GS_WPE) k_wire_fq(size_t n, int dir, const uint8_t* in, uint8_t* out, uint8_t* ok) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  constexpr int B = C::N * 4;
  uint32_t w[C::N];
  if (dir == 0) {
    BFq<C> b = reinterpret_cast<const BFq<C>*>(in)[g];
    fq_to_canonical<C>(w, fq_from_boundary<C>(b.w));
    for (int i = 0; i < B; i++) out[g * B + i] = (uint8_t)(w[i >> 2] >> ((i & 3) * 8));
  } else {
    for (int i = 0; i < C::N; i++) w[i] = 0;
    for (int i = 0; i < B; i++) w[i >> 2] |= (uint32_t)in[g * B + i] << ((i & 3) * 8);
    bool canon = words_lt_p<C>(w);
    BFq<C> b;
    if (!canon)
      for (int i = 0; i < C::N; i++) w[i] = 0;
    fq_to_boundary<C>(b.w, fq_from_canonical<C>(w));
    reinterpret_cast<BFq<C>*>(out)[g] = b;
    ok[g] = canon ? 1 : 0;
  }
}
template <class C>
__global__ void __launch_bounds__(64, GS_WPE) k_wire_fr(size_t n, int dir, const uint8_t* in, uint8_t* out, uint8_t* ok) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  constexpr int NW = FrM<C>::N, B = NW * 4;
  if (dir == 0) {
    Fr<C> k = from_mont(reinterpret_cast<const Fr<C>*>(in)[g]);
    for (int i = 0; i < B; i++) out[g * B + i] = (uint8_t)(k.v[i >> 2] >> ((i & 3) * 8));
  } else {
    Fr<C> k;
    uint32_t r[NW];
    for (int i = 0; i < NW; i++) {
      k.v[i] = 0;
      r[i] = C::R_WORDS[i];
    }
    for (int i = 0; i < B; i++) k.v[i >> 2] |= (uint32_t)in[g * B + i] << ((i & 3) * 8);
    bool canon = words_gt<NW>(r, k.v);
    if (!canon)
      for (int i = 0; i < NW; i++) k.v[i] = 0;
    reinterpret_cast<Fr<C>*>(out)[g] = to_mont(k);
    ok[g] = canon ? 1 : 0;
  }
}
template <class C> __global__ void __launch_bounds__(64, GS_WPE) k_wire_gt_check(size_t n, const uint8_t* gt, uint8_t* ok) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  Fp12<C> f;
  f12_from_boundary<C>(f, reinterpret_cast<const BFq<C>*>(gt) + 12 * g);
  ok[g] = (ok[g] && f12_in_torsion(f)) ? 1 : 0;
}

}  // namespace gs
