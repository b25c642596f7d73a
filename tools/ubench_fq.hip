// Micro-benchmark of the field / tower code at different wave residencies (not part of the product):
//   how many Fq multiplications per SIMD-cycle do (a) a bare chain of mul28 calls and (b) the Miller inner loop
//   (f12_sqr + 2 sparse line products) reach with 1, 2, 4 waves per SIMD, and what does a 256-register build of (b)
//   cost?  Build (repo root) and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 --gpu-max-threads-per-block=64 -DGS_F6_INLINE -DGS_JAC_INLINE \
//         -Wno-unused-value -Wno-psabi tools/ubench_fq.hip -o tools/ubench_fq && tools/ubench_fq
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../groth_sahai_rs_amd/csrc/gs_params_bls12_381.h"
#include "../groth_sahai_rs_amd/csrc/gs_pairing.cuh"
#include "../groth_sahai_rs_amd/csrc/gs_coop.cuh"

namespace gs {
GS_ZERO_ONE(Bls12_381)
}
using namespace gs;
typedef Bls12_381 C;

__device__ Fq<C> seed_fq(uint32_t s) {
  Fq<C> r;
  for (int i = 0; i < C::L; i++) r.v[i] = (int32_t)((s * 2654435761u + i * 40503u) & 0x0FFFFFFF);
  r.v[C::L - 1] &= 0xFFFF;
  return r;
}
__device__ Fp2<C> seed_fp2(uint32_t s) { return {seed_fq(s), seed_fq(s * 7 + 1)}; }

// (a) bare chain: 2 multiplications + 1 squaring per iteration
#ifndef UB_BLOCK
#define UB_BLOCK 64
#endif
__global__ void __launch_bounds__(UB_BLOCK) k_chain(uint32_t* out, int iters) {
  Fq<C> x = seed_fq(threadIdx.x % 64 + 1), y = seed_fq(threadIdx.x % 64 + 77);
  for (int i = 0; i < iters; i++) {
    x = mul(x, y);
    y = sqr(y);
    y = mul(y, x);
  }
  out[blockIdx.x * UB_BLOCK + threadIdx.x] = (uint32_t)(x.v[0] + y.v[3]);
}

// (b) Miller inner loop on one accumulator: square, two sparse line products (36 + 2 x 39 = 114 Fq multiplications)
template <int WPE> __device__ __forceinline__ void miller_body(uint32_t* out, int iters) {
  Fp12<C> f;
  f12_one(f);
  f.c0.c1 = seed_fp2(threadIdx.x + 3);
  f.c1.c2 = seed_fp2(threadIdx.x + 5);
  Fp2<C> l0 = seed_fp2(threadIdx.x + 11), l1 = seed_fp2(threadIdx.x + 13), l4 = seed_fp2(threadIdx.x + 17);
  for (int i = 0; i < iters; i++) {
    f12_sqr(f, f);
    f12_mul_by_014(f, l0, l1, l4);
    f12_mul_by_014(f, l1, l4, l0);
  }
  out[blockIdx.x * UB_BLOCK + threadIdx.x] = (uint32_t)(f.c0.c0.c0.v[0] + f.c1.c1.c1.v[2]);
}
#ifndef UB_WPE
#define UB_WPE 1
#endif
__global__ void __launch_bounds__(UB_BLOCK, UB_WPE) k_miller_w1(uint32_t* out, int iters) { miller_body<1>(out, iters); }

// (c) the existing 3-lane cooperative steps (gs_coop.cuh): 3 Granger-Scott squarings + 1 product by a base per
// iteration = 3 x 6 + 27 = 45 Fq multiplications per LANE; 21 groups per wave
__global__ void __launch_bounds__(UB_BLOCK, UB_WPE) k_coop(uint32_t* out, int iters) {
  int lane = threadIdx.x % 64, j = lane % 3;
  if (lane >= 63) return;
  Fp4<C> acc = {seed_fp2(lane + 3), seed_fp2(lane + 5)};
  Fp4<C> Bs[3];
  for (int i = 0; i < 3; i++) Bs[i] = {seed_fp2(lane + 7 + i), seed_fp2(lane + 17 + i)};
  CoopWave xw{lane - j};
  for (int i = 0; i < iters; i++) {
    c12_sqr_step(acc, j, xw);
    c12_sqr_step(acc, j, xw);
    c12_sqr_step(acc, j, xw);
    c12_mul_step(acc, Bs, j, xw);
  }
  out[blockIdx.x * UB_BLOCK + threadIdx.x] = (uint32_t)(acc.a.c0.v[0] + acc.b.c1.v[2]);
}

template <class K> void run(const char* name, K kern, int muls_per_iter, int iters, uint32_t* d) {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  int simds = pr.multiProcessorCount * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-34s", name);
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = simds * wps * 64 / UB_BLOCK;  // UB_BLOCK = 256: one wave of a block per SIMD of its CU (balanced)
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(UB_BLOCK), 0, 0, d, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(UB_BLOCK), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_muls_per_simd = (double)wps * iters * muls_per_iter;
    double us_per = ms * 1e3 / wave_muls_per_simd;
    printf("  w%d: %7.1f ms  %6.3f us/wave-mul/SIMD (%5.0f cyc @2.1GHz)", wps, ms, us_per, us_per * 2100.0);
  }
  printf("\n");
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 64u * 4096 * 16);
  run("mul chain (2 mul + 1 sqr)", k_chain, 3, 20000, d);
  run(UB_WPE == 1 ? "miller body, 512 regs" : "miller body, 256 regs", k_miller_w1, 114, 400, d);
  run(UB_WPE == 1 ? "coop steps (45/lane), 512 regs" : "coop steps (45/lane), 256 regs", k_coop, 45, 1000, d);
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
