// Micro-benchmark of the VALU instructions that bound multi-precision modular
// arithmetic on gfx950: throughput in wave-instructions per SIMD-cycle at 1..8
// waves/SIMD.  Not part of the product; run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP> __global__ void kern(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
  uint32_t b = seed | 1, c = seed * 77 + 5;
  uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  double f0 = a0, f1 = a1, f2 = a2, f3 = a3, fb = 1.0000001, fc = 0.5;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) {  // v_mad_u64_u32, 4 independent chains
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 1) {  // v_mul_lo_u32
      REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if (OP == 2) {  // v_mul_hi_u32
      REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if (OP == 3) {  // v_mad_u32_u24
      REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    } else if (OP == 4) {  // v_fma_f64
      REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));)
    } else if (OP == 5) {  // v_add_co_u32 + v_addc_co_u32 pair (count as 2)
      REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
    } else if (OP == 6) {  // v_lshl_add_u64
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d3));)
    } else if (OP == 7) {  // v_mad_u64_u32 + v_addc (Comba step), 2 independent accumulators
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(d0), "+v"(d1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 8) {  // v_mul_u32_u24 + v_mul_hi_u32_u24
      REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if (OP == 10) {  // ONE dependent chain: mad -> addc -> mad ... (the Comba inner step)
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(d0), "+v"(a2) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 11) {  // ONE dependent chain of v_mad_u64_u32 only
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(b), "v"(c) : "vcc");)
    } else if (OP == 12) {  // ONE dependent chain of v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));)
    } else if (OP == 9) {  // v_add_u32 (full-rate reference)
      REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(d0 + d1 + d2 + d3) + (uint32_t)(f0 + f1 + f2 + f3);
}

template <int OP> void run(const char* name, uint32_t* d) {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  int cus = pr.multiProcessorCount;
  double clk = pr.clockRate * 1e3;  // Hz
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-28s", name);
  for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
    int threads = 256, blocks = cus * wps;  // 4 waves per block -> one per SIMD
    int iters = 2000;
    hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)iters * 64.0 * wps;        // wave-instructions per SIMD
    double cyc = ms * 1e-3 * clk;
    printf("  w%d: %6.2f cyc/inst", wps, cyc / winst);
  }
  printf("\n");
}

__global__ void clk_probe(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t a = threadIdx.x, b = 12345;
  for (int i = 0; i < iters; i++) { REP16(asm volatile("v_mad_u64_u32 v[100:101], vcc, %0, %1, v[100:101]" :: "v"(a), "v"(b) : "vcc", "v100", "v101");) }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
int main() {
  {
    unsigned long long* o; hipMalloc(&o, 16);
    hipLaunchKernelGGL(clk_probe, dim3(256 * 8), dim3(256), 0, 0, o, 200000);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("in-kernel clock under integer load: %.3f GHz (s_memtime %llu / s_memrealtime %llu x 100 MHz)\n", (double)h[0] / (double)h[1] * 0.1, h[0], h[1]);
  }
  uint32_t* d;
  hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  printf("device %s, %d CUs, clockRate %d kHz (cycles computed at that clock)\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  run<9>("v_add_u32", d);
  run<0>("v_mad_u64_u32", d);
  run<7>("v_mad_u64_u32+v_addc (x2)", d);
  run<1>("v_mul_lo_u32", d);
  run<2>("v_mul_hi_u32", d);
  run<3>("v_mad_u32_u24", d);
  run<8>("v_mul(_hi)_u32_u24", d);
  run<4>("v_fma_f64", d);
  run<5>("v_add_co/v_addc_co", d);
  run<6>("v_lshl_add_u64", d);
  run<10>("dep chain mad+addc", d);
  run<11>("dep chain mad only", d);
  run<12>("dep chain v_add_u32", d);
  return 0;
}
