// Micro-benchmark of the VALU instructions that bound multi-precision modular
// arithmetic on gfx950: cost in cycles per wave-instruction per SIMD at 1..8
// waves/SIMD.  Not part of the product; run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench && tools/ubench
//
// Round 4: clock and rate come from ONE measurement.  Every wave stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz) around its own instruction stream; a row prints
//   clk   the clock THAT kernel ran at  (median over waves of d memtime / d memrealtime x 100 MHz)
//   cyc   REAL shader cycles per wave-instruction per SIMD  (median d memtime / (instructions x waves per SIMD))
//   wall  the same from the launch's wall time at the nominal clockRate (what rounds 1-3 printed)
// so that peak = SIMDs x 64 / cyc x clk needs no second clock correction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

struct Stamp {
  unsigned long long t0, t1, r0, r1;
};

template <int OP> __global__ void kern(uint32_t* out, Stamp* st, int iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
  uint32_t b = seed | 1, c = seed * 77 + 5;
  uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  double f0 = a0, f1 = a1, f2 = a2, f3 = a3, fb = 1.0000001, fc = 0.5;
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i q0 = {(int)a0, (int)a1, (int)a2, (int)a3}, q1 = {(int)a3, (int)a2, (int)a1, (int)a0};
  __shared__ int4 lds[512];
  lds[threadIdx.x] = make_int4(a0, a1, a2, a3);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) {  // v_mad_u64_u32, 4 independent chains
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 1) {  // v_mul_lo_u32
      REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if (OP == 2) {  // v_mad_i64_i32 (the product half of the multiplier)
      REP16(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 3) {  // the multiplier's mix: 7 mads per full-rate op, ONE accumulator chain pair (as gs_fp2mul28_sub)
      REP16(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %0, vcc, %5, %4, %0\n v_mad_i64_i32 %1, vcc, %5, %4, %1\n"
                         "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %0, vcc, %5, %4, %0\n v_and_b32 %2, 0xfffffff, %3" : "+v"(d0), "+v"(d1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 4) {  // v_fma_f64
      REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));)
    } else if (OP == 5) {  // v_add_co_u32 + v_addc_co_u32 pair (count as 2)
      REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
    } else if (OP == 6) {  // v_lshl_add_u64
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d3));)
    } else if (OP == 7) {  // carry round of the tower code: and, ashr, add (3 full-rate ops per limb), 4 ops per group
      REP16(asm volatile("v_and_b32 %0, 0xfffffff, %1\n v_ashrrev_i32 %2, 28, %3\n v_add_u32 %0, %0, %2\n v_sub_u32 %1, %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 8) {  // v_accvgpr_write + v_accvgpr_read (the register-file extension traffic), 2 + 2
      REP16(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_read_b32 %2, a0\n v_accvgpr_read_b32 %3, a1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "a0", "a1");)
    } else if (OP == 10) {  // v_mov_b32_dpp quad_perm [1,0,3,2] (the lane-pair exchange)
      REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (OP == 11) {  // ONE dependent chain of v_mad_u64_u32 only
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 12) {  // ONE dependent chain of v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));)
    } else if (OP == 13) {  // ds_write_b128 + ds_read_b128 to the lane's own slot (the line exchange), 2 + 2
      REP16(asm volatile("ds_write_b128 %2, %0\n ds_read_b128 %0, %2\n ds_write_b128 %2, %1 offset:4096\n ds_read_b128 %1, %2 offset:4096\n s_waitcnt lgkmcnt(0)"
                         : "+v"(q0), "+v"(q1) : "v"((uint32_t)(threadIdx.x * 16)) : "memory");)
    } else if (OP == 9) {  // v_add_u32 (full-rate reference)
      REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    Stamp s{t0, t1, r0, r1};
    st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(d0 + d1 + d2 + d3) + (uint32_t)(f0 + f1 + f2 + f3) + lds[(threadIdx.x + 1) & 255].x + q0.x + q1.y;
}

static double median(std::vector<double>& v) {
  std::sort(v.begin(), v.end());
  return v.empty() ? 0.0 : v[v.size() / 2];
}

template <int OP> void run(const char* name, uint32_t* d, Stamp* st, int per_iter = 64) {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  int cus = pr.multiProcessorCount;
  double clk = pr.clockRate * 1e3;  // Hz
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-30s", name);
  for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
    int threads = 256, blocks = cus * wps;  // 4 waves per block -> one per SIMD
    int iters = 20000;                      // ~10 ms per launch at one wave: long enough for the clock to settle
    hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(threads), 0, 0, d, st, iters / 4, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(threads), 0, 0, d, st, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int nw = blocks * threads / 64;
    std::vector<Stamp> h(nw);
    hipMemcpy(h.data(), st, nw * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> ck, cy;
    double winst = (double)iters * per_iter;  // wave-instructions of ONE wave
    for (auto& s : h) {
      if (s.r1 == s.r0) continue;
      ck.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
      cy.push_back((double)(s.t1 - s.t0) / (winst * wps));
    }
    printf("  w%d: clk %.3f cyc %5.2f wall %5.2f", wps, median(ck), median(cy), ms * 1e-3 * clk / (winst * wps));
  }
  printf("\n");
}

int main() {
  uint32_t* d;
  Stamp* st;
  hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
  hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  printf("device %s, %d CUs, clockRate %d kHz; per row and occupancy: clk = GHz of THAT kernel (s_memtime / s_memrealtime),\n"
         "cyc = real shader cycles per wave-instruction per SIMD (s_memtime), wall = wall time x clockRate (rounds 1-3)\n",
         pr.name, pr.multiProcessorCount, pr.clockRate);
  run<9>("v_add_u32", d, st);
  run<0>("v_mad_u64_u32", d, st);
  run<2>("v_mad_i64_i32", d, st);
  run<3>("multiplier mix (7 mad : 1 and)", d, st, 128);
  run<1>("v_mul_lo_u32", d, st);
  run<4>("v_fma_f64", d, st);
  run<5>("v_add_co/v_addc_co", d, st);
  run<6>("v_lshl_add_u64", d, st);
  run<7>("carry round and/ashr/add/sub", d, st);
  run<8>("v_accvgpr_write/read", d, st);
  run<10>("v_mov_b32_dpp (+s_nop 1 per 3)", d, st);
  run<13>("ds_write/read_b128 +wait", d, st, 80);
  run<11>("dep chain mad only", d, st);
  run<12>("dep chain v_add_u32", d, st);
  return 0;
}
