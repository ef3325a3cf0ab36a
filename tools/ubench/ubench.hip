// Microbenchmarks that size the integer-VALU roofline of the path ("the integer MAD rate is not in the
// guides, must be measured"): cycles per wave-instruction for the candidate multiply primitives at 1/2/4 waves per
// SIMD, including the carry-chain forms (v_addc_co_u32 with s_nop padding) that decided the field representation.
// Fp-multiplication throughput of the chosen 28-bit-limb form: tools/ubench/ubench28.hip.  (The numbers of the earlier
// saturated 12 x 32-bit multiplier, 58 G fp_mul/s, are kept in profiles/ubench_r01.txt.)
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define ITERS 512

enum { I_ADD32, I_MAD64, I_MULLO, I_MULHI, I_MAD24, I_LSHLADD64, I_FMA64, I_FMA32, I_ADDC, I_MADCARRY, I_CHAIN_NOP, I_CHAIN2_NOP0, I_N };
static const char* NAMES[] = {"v_add_u32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_lshl_add_u64",
                              "v_fma_f64", "v_fma_f32", "v_addc_co_u32(3 chains)", "mad_u64+addc(grp3)", "addc 1 chain + s_nop 1", "addc 2 chains + s_nop 0"};

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint64_t* cycles, uint32_t* sink, uint32_t seed) {
  uint32_t x0 = threadIdx.x + seed, x1 = x0 * 3 + 1, x2 = x0 * 5 + 7, x3 = x0 * 7 + 3;
  uint64_t a0 = x0, a1 = x1, a2 = x2, a3 = x3, a4 = x0 + 9, a5 = x1 + 9, a6 = x2 + 9, a7 = x3 + 9;
  double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = 1.5, d5 = 2.5, d6 = 3.5, d7 = 4.5;
  float f0 = x0, f1 = x1, f2 = x2, f3 = x3, f4 = 1.5f, f5 = 2.5f, f6 = 3.5f, f7 = 4.5f;
  uint32_t w0 = x0, w1 = x1, w2 = x2, w3 = x3, w4 = x0 ^ 5, w5 = x1 ^ 5, w6 = x2 ^ 5, w7 = x3 ^ 5;
  uint64_t s0 = 0, s1 = 0, s2 = 0;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
    if (KIND == I_ADD32) {
      REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(x0));)
    } else if (KIND == I_MAD64) {
      REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n"
                        "v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
                        "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x0), "v"(x1) : "vcc");)
    } else if (KIND == I_MULLO) {
      REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                        "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(x1));)
    } else if (KIND == I_MULHI) {
      REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                        "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8"
                        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(x1));)
    } else if (KIND == I_MAD24) {
      REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n"
                        "v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7"
                        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(x1));)
    } else if (KIND == I_LSHLADD64) {
      REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                        "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));)
    } else if (KIND == I_FMA64) {
      REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                        "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(d0), "v"(d1));)
    } else if (KIND == I_FMA32) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(f0), "v"(f1));)
    } else if (KIND == I_ADDC) {
      // three interleaved carry chains on three SGPR pairs: 2 wait states between dependent links, no s_nop needed
      REP8(asm volatile("v_addc_co_u32 %0, %3, %0, %6, %3\n v_addc_co_u32 %1, %4, %1, %6, %4\n v_addc_co_u32 %2, %5, %2, %6, %5\n"
                        "v_addc_co_u32 %0, %3, %0, %6, %3\n v_addc_co_u32 %1, %4, %1, %6, %4\n v_addc_co_u32 %2, %5, %2, %6, %5\n"
                        "v_addc_co_u32 %0, %3, %0, %6, %3\n v_addc_co_u32 %1, %4, %1, %6, %4"
                        : "+v"(w0), "+v"(w1), "+v"(w2), "+s"(s0), "+s"(s1), "+s"(s2) : "v"(x1));)
    } else if (KIND == I_CHAIN_NOP) {
      // what hipcc emits for a 12-limb add: every link followed by s_nop 1 (8 links counted per asm)
      REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1\n"
                        "v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1\n"
                        "v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1\n"
                        "v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n s_nop 1"
                        : "+v"(w0) : "v"(x1) : "vcc");)
    } else if (KIND == I_CHAIN2_NOP0) {
      REP8(asm volatile("v_addc_co_u32 %0, %2, %0, %4, %2\n v_addc_co_u32 %1, %3, %1, %4, %3\n s_nop 0\n"
                        "v_addc_co_u32 %0, %2, %0, %4, %2\n v_addc_co_u32 %1, %3, %1, %4, %3\n s_nop 0\n"
                        "v_addc_co_u32 %0, %2, %0, %4, %2\n v_addc_co_u32 %1, %3, %1, %4, %3\n s_nop 0\n"
                        "v_addc_co_u32 %0, %2, %0, %4, %2\n v_addc_co_u32 %1, %3, %1, %4, %3\n s_nop 0"
                        : "+v"(w0), "+v"(w1), "+s"(s0), "+s"(s1) : "v"(x1));)
    } else if (KIND == I_MADCARRY) {
      // the fp_mul inner pattern: 3 mads then 3 addc (6 instructions)
      REP8(asm volatile("v_mad_u64_u32 %0, %2, %5, %6, %0\n v_mad_u64_u32 %0, %3, %5, %7, %0\n v_mad_u64_u32 %0, %4, %6, %7, %0\n"
                        "v_addc_co_u32 %1, %2, 0, %1, %2\n v_addc_co_u32 %1, %3, 0, %1, %3\n v_addc_co_u32 %1, %4, 0, %1, %4"
                        : "+v"(a0), "+v"(w0), "=&s"(s0), "=&s"(s1), "=&s"(s2) : "v"(x1), "v"(x2), "v"(x3));)
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t r = w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7 ^ (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^
               (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
  if (r == 0x12345) sink[0] = r;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static double median_cycles(std::vector<uint64_t>& v) {
  std::sort(v.begin(), v.end());
  return (double)v[v.size() / 2];
}

template <int KIND>
static void run_rate(int wps, uint64_t* dcy, uint32_t* dsink) {
  int blocks = 256 * wps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_rate<KIND><<<blocks, 256>>>(dcy, dsink, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_rate<KIND><<<blocks, 256>>>(dcy, dsink, 2);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint64_t> cy(blocks * 4);
  CK(hipMemcpy(cy.data(), dcy, cy.size() * 8, hipMemcpyDeviceToHost));
  double med = median_cycles(cy);
  int per_iter = (KIND == I_MADCARRY) ? 48 : 64;
  double n_inst = (double)ITERS * per_iter;
  // cycles per wave-instruction as seen by one wave; x waves-per-SIMD sharing the SIMD => SIMD issue interval = that / wps
  printf("%-26s wps=%d  cyc/inst(wave)=%7.2f  SIMD-interval=%6.2f cyc  wall=%.3f ms  chip rate=%.2f T lane-op/s\n", NAMES[KIND], wps,
         med / n_inst, med / n_inst / wps, ms, (double)blocks * 256 * n_inst / (ms * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  uint64_t* dcy; uint32_t* dsink;
  CK(hipMalloc(&dcy, 8 * 4 * 256 * 8)); CK(hipMalloc(&dsink, 64));
  for (int wps = 1; wps <= 4; wps *= 2) {
    run_rate<I_ADD32>(wps, dcy, dsink); run_rate<I_MAD64>(wps, dcy, dsink); run_rate<I_MULLO>(wps, dcy, dsink);
    run_rate<I_MULHI>(wps, dcy, dsink); run_rate<I_MAD24>(wps, dcy, dsink); run_rate<I_LSHLADD64>(wps, dcy, dsink);
    run_rate<I_FMA64>(wps, dcy, dsink); run_rate<I_FMA32>(wps, dcy, dsink); run_rate<I_ADDC>(wps, dcy, dsink);
    run_rate<I_MADCARRY>(wps, dcy, dsink); run_rate<I_CHAIN_NOP>(wps, dcy, dsink); run_rate<I_CHAIN2_NOP0>(wps, dcy, dsink);
  }
  return 0;
}
