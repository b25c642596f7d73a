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

int main() {
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
  return 0;
}
