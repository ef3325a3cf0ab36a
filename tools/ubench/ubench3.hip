// Round-3 microbenchmarks: what bounds the 28-bit-limb multiplier leaf?  (profiles/ubench_r01_rates.txt: v_mad_u64_u32 on two
// fixed sources issues every 3.3 cycles per SIMD at four waves, but the real leaf runs at ~4.6 cycles per instruction at two AND
// at four waves per SIMD.)  Part A: v_mad_i64_i32 with realistic operand patterns (distinct sources, one dependent accumulator
// chain vs several, an SGPR source) and the other 64-bit instructions of the leaf, at 1..8 waves per SIMD.  Part B: three
// formulations of the fused two-product Montgomery pass REDC(a b + c d), at 1..4 waves per SIMD, each checked against the host.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench3 tools/ubench/ubench3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#if defined(__HIP_DEVICE_COMPILE__)
#define MADV(acc, a, b) do { uint64_t cy_; asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy_) : "v"(a), "v"(b)); } while (0)
#define MADS(acc, a, b) do { uint64_t cy_; asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy_) : "v"(a), "s"(b)); } while (0)
#define ASHR(acc) asm volatile("v_ashrrev_i64 %0, 28, %0" : "+v"(acc))
#define ADD32(x, y) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y))
#else
#define MADV(acc, a, b) acc += (int64_t)(a) * (b)
#define MADS(acc, a, b) acc += (int64_t)(a) * (b)
#define ASHR(acc) acc >>= 28
#define ADD32(x, y) x += y
#endif

enum { A_8ACC_SAME, A_8ACC_DIST, A_1ACC_DIST, A_2ACC_DIST, A_4ACC_DIST, A_8ACC_SGPR, A_ASHR64, A_MAD3_ADD1, A_N };
static const char* ANAMES[] = {"mad_i64 8 acc, same src", "mad_i64 8 acc, distinct src", "mad_i64 1 acc (chain), distinct", "mad_i64 2 acc, distinct",
                               "mad_i64 4 acc, distinct", "mad_i64 8 acc, sgpr src1", "v_ashrrev_i64 8 chains", "3 mad (1 chain) + 1 v_add"};
#define ITERS 256

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint64_t* cycles, uint32_t* sink, uint32_t seed, int sval) {
  int32_t a[8], b[8];
  int64_t acc[8];
  uint32_t w[4];
  for (int i = 0; i < 8; i++) {
    a[i] = (int32_t)(threadIdx.x * (2 * i + 3) + seed);
    b[i] = (int32_t)(threadIdx.x * (2 * i + 5) + 7 * seed);
    acc[i] = i + seed;
  }
  for (int i = 0; i < 4; i++) w[i] = threadIdx.x + i;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int n = 8 * i + j;
        if (KIND == A_8ACC_SAME) MADV(acc[n & 7], a[0], b[0]);
        else if (KIND == A_8ACC_DIST) MADV(acc[n & 7], a[i], b[j]);
        else if (KIND == A_1ACC_DIST) MADV(acc[0], a[i], b[j]);
        else if (KIND == A_2ACC_DIST) MADV(acc[n & 1], a[i], b[j]);
        else if (KIND == A_4ACC_DIST) MADV(acc[n & 3], a[i], b[j]);
        else if (KIND == A_8ACC_SGPR) MADS(acc[n & 7], a[i], sval + j);
        else if (KIND == A_ASHR64) ASHR(acc[n & 7]);
        else if (KIND == A_MAD3_ADD1) {
          if ((n & 3) == 3) ADD32(w[(n >> 2) & 3], a[i]);
          else MADV(acc[0], a[i], b[j]);
        }
      }
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t r = 0;
  for (int i = 0; i < 8; i++) r ^= (uint64_t)acc[i];
  for (int i = 0; i < 4; i++) r ^= w[i];
  if (r == 0x12345) sink[0] = (uint32_t)r;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run_rate(int wps, uint64_t* dcy, uint32_t* dsink) {
  int blocks = 256 * wps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_rate<KIND><<<blocks, 256>>>(dcy, dsink, 1, 3);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_rate<KIND><<<blocks, 256>>>(dcy, dsink, 2, 3);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint64_t> cy(blocks * 4);
  CK(hipMemcpy(cy.data(), dcy, cy.size() * 8, hipMemcpyDeviceToHost));
  std::sort(cy.begin(), cy.end());
  double med = (double)cy[cy.size() / 2], n_inst = (double)ITERS * 64;
  printf("%-34s wps=%d  cyc/inst(wave)=%7.2f  SIMD-interval=%6.2f cyc  wall=%.3f ms  chip rate=%.2f T lane-op/s\n", ANAMES[KIND], wps, med / n_inst,
         med / n_inst / wps, ms, (double)blocks * 256 * n_inst / (ms * 1e-3) / 1e12);
}

// ---------------------------------------------------------------- part B: the fused pass in three formulations
#define NL 14
#define MASK28 0x0fffffff
#define N0INV28 0xffcfffdu
#define HD __host__ __device__ __forceinline__
struct fp28 { int32_t l[NL]; };
__device__ __constant__ const int32_t P28_D[NL] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};
static const int32_t P28_H[NL] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};

#if defined(__HIP_DEVICE_COMPILE__)
#define OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define OPAQUE(x)
#endif

// F = 0: product scanning, one accumulator (the library's form, fp.cuh fp_redc_products<2>)
// F = 1: product scanning, the two product streams and the reduction stream in three accumulators, summed at the column's end
// F = 2: operand scanning: 15 independent 64-bit column accumulators, row i adds a_i b + c_i d + m_i p
// F = 3: product scanning with the products of column k + 1 issued in a second accumulator while column k's reduction chain runs
#if defined(__HIP_DEVICE_COMPILE__)
// F = 4: one accumulator chain, every multiply-add an opaque instruction (the compiler cannot split the chain into several
// accumulators that it then joins with 64-bit additions), the reduction's products unsigned
#define MADI(acc, a, b) do { uint64_t cy_; asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy_) : "v"(a), "v"(b)); } while (0)
#define MADU(acc, a, b) do { uint64_t cy_; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy_) : "v"(a), "s"(b)); } while (0)
__device__ __forceinline__ void fused_asm(fp28& r, const fp28& a, const fp28& b, const fp28& c, const fp28& d, const int32_t* P) {
  int64_t acc = 0;
  int32_t m[NL];
  fp28 t;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    const int lo = k > NL - 1 ? k - (NL - 1) : 0, hi = k < NL - 1 ? k : NL - 1;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      MADI(acc, a.l[i], b.l[k - i]);
      MADI(acc, c.l[i], d.l[k - i]);
    }
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (k < NL && i == k) continue;
      MADU(acc, m[i], P[k - i]);
    }
    if (k < NL) {
      m[k] = (int32_t)(((uint32_t)acc * N0INV28) & MASK28);
      MADU(acc, m[k], P[0]);
      acc >>= 28;
    } else {
      t.l[k - NL] = (int32_t)((uint32_t)acc & MASK28);
      acc >>= 28;
    }
  }
  t.l[NL - 1] = (int32_t)acc;
  r = t;
}
#endif
template <int F>
HD void fused(fp28& r, const fp28& a_, const fp28& b_, const fp28& c_, const fp28& d_, const int32_t* P) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (F == 4) { fused_asm(r, a_, b_, c_, d_, P); return; }
#endif
  fp28 a = a_, b = b_, c = c_, d = d_;
  for (int i = 0; i < NL; i++) { OPAQUE(a.l[i]); OPAQUE(b.l[i]); OPAQUE(c.l[i]); OPAQUE(d.l[i]); }
  fp28 t;
  if (F == 0 || F == 1 || F == 3) {
    int64_t acc = 0;
    int32_t m[NL];
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      const int lo = k > NL - 1 ? k - (NL - 1) : 0, hi = k < NL - 1 ? k : NL - 1;
      int64_t p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
      for (int i = lo; i <= hi; i++) {
        if (F == 0) { acc += (int64_t)a.l[i] * b.l[k - i]; acc += (int64_t)c.l[i] * d.l[k - i]; }
        else { p1 += (int64_t)a.l[i] * b.l[k - i]; p2 += (int64_t)c.l[i] * d.l[k - i]; }
      }
#pragma unroll
      for (int i = lo; i <= hi; i++) {
        if (k < NL && i == k) continue;
        if (F == 0) acc += (int64_t)m[i] * P[k - i];
        else p3 += (int64_t)m[i] * P[k - i];
      }
      if (F != 0) acc += p1 + p2 + p3;
      if (k < NL) {
        m[k] = (int32_t)(((uint32_t)acc * N0INV28) & MASK28);
        acc += (int64_t)m[k] * P[0];
        acc >>= 28;
      } else {
        t.l[k - NL] = (int32_t)((uint32_t)acc & MASK28);
        acc >>= 28;
      }
    }
    t.l[NL - 1] = (int32_t)acc;
  } else {
    int64_t col[NL + 1];
#pragma unroll
    for (int j = 0; j <= NL; j++) col[j] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
      for (int j = 0; j < NL; j++) {
        col[j] += (int64_t)a.l[i] * b.l[j];
        col[j] += (int64_t)c.l[i] * d.l[j];
      }
      const int32_t m = (int32_t)(((uint32_t)col[0] * N0INV28) & MASK28);
#pragma unroll
      for (int j = 0; j < NL; j++) col[j] += (int64_t)m * P[j];
      const int64_t carry = col[0] >> 28;
#pragma unroll
      for (int j = 0; j < NL; j++) col[j] = col[j + 1];
      col[0] += carry;
      col[NL] = 0;
    }
    int64_t cy = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      cy += col[j];
      if (j < NL - 1) { t.l[j] = (int32_t)((uint32_t)cy & MASK28); cy >>= 28; }
      else t.l[j] = (int32_t)cy;
    }
  }
  r = t;
}

template <int F, int LB>
__global__ void __launch_bounds__(256, LB) k_mul(int32_t* out, const int32_t* in, int iters, uint64_t* cycles) {
  int id = blockIdx.x * blockDim.x + threadIdx.x;
  fp28 x, y, z;
  for (int i = 0; i < NL; i++) x.l[i] = in[NL * (id & 1023) + i];
  for (int i = 0; i < NL; i++) y.l[i] = in[NL * ((id + 3) & 1023) + i];
  for (int i = 0; i < NL; i++) z.l[i] = in[NL * ((id + 5) & 1023) + i];
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) fused<F>(x, x, y, z, x, P28_D);
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < NL; i++) out[NL * id + i] = x.l[i];
  if ((threadIdx.x & 63) == 0) cycles[id >> 6] = t1 - t0;
}

template <int F, int LB>
static void run_mul(int wps, int32_t* dout, int32_t* din, uint64_t* dcy, const std::vector<int32_t>& hin, int iters = 200) {
  int blocks = 256 * wps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_mul<F, LB><<<blocks, 256>>>(dout, din, iters, dcy);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_mul<F, LB><<<blocks, 256>>>(dout, din, iters, dcy);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<int32_t> o(NL * 1024);
  CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint64_t> cy(blocks * 4);
  CK(hipMemcpy(cy.data(), dcy, cy.size() * 8, hipMemcpyDeviceToHost));
  std::sort(cy.begin(), cy.end());
  bool ok = true;
  for (int id = 0; id < 64 && ok; id++) {
    fp28 x, y, z;
    for (int i = 0; i < NL; i++) x.l[i] = hin[NL * (id & 1023) + i];
    for (int i = 0; i < NL; i++) y.l[i] = hin[NL * ((id + 3) & 1023) + i];
    for (int i = 0; i < NL; i++) z.l[i] = hin[NL * ((id + 5) & 1023) + i];
    if (iters > 1000) break;            // long power / clock runs: no host replay
    for (int it = 0; it < iters; it++) fused<0>(x, x, y, z, x, P28_H);
    for (int i = 0; i < NL; i++) if (x.l[i] != o[NL * id + i]) ok = false;
  }
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)k_mul<F, LB>));
  printf("fused form=%d lb=%d wps=%d  regs=%d scratch=%zu  wall=%.3f ms  %.2f G pass/s = %.2f G fp_mul-equiv/s  wave cycles/pass=%.0f  %s\n", F, LB, wps, fa.numRegs,
         (size_t)fa.localSizeBytes, ms, (double)blocks * 256 * iters / (ms * 1e-3) / 1e9, (double)blocks * 256 * iters * 1.5 / (ms * 1e-3) / 1e9,
         (double)cy[cy.size() / 2] / iters, ok ? "[matches host]" : "[MISMATCH vs host]");
}

template <int K>
static void all_rates(uint64_t* dcy, uint32_t* dsink) {
  for (int wps = 1; wps <= 8; wps *= 2) run_rate<K>(wps, dcy, dsink);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  uint64_t* dcy; uint32_t* dsink; int32_t *din, *dout;
  CK(hipMalloc(&dcy, 8 * 4 * 256 * 8)); CK(hipMalloc(&dsink, 64));
  CK(hipMalloc(&din, NL * 4 * 1024)); CK(hipMalloc(&dout, NL * 4 * 256 * 256 * 8));
  if (argc > 2 && !strcmp(argv[1], "long")) {   // ubench3 long <wps>: ~2 s of the fused pass at that occupancy, for rocm-smi power / clock sampling
    std::vector<int32_t> hin(NL * 1024, 12345);
    CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    const int wps = atoi(argv[2]);
    for (int rep = 0; rep < 3; rep++) {
      if (wps == 1) run_mul<0, 1>(1, dout, din, dcy, hin, 300000);
      else if (wps == 2) run_mul<0, 4>(2, dout, din, dcy, hin, 150000);
      else run_mul<0, 4>(4, dout, din, dcy, hin, 75000);
    }
    return 0;
  }
  const bool leaf_only = argc > 1 && !strcmp(argv[1], "leaf");
  if (!leaf_only) {
  all_rates<A_8ACC_SAME>(dcy, dsink);
  all_rates<A_8ACC_DIST>(dcy, dsink);
  all_rates<A_1ACC_DIST>(dcy, dsink);
  all_rates<A_2ACC_DIST>(dcy, dsink);
  all_rates<A_4ACC_DIST>(dcy, dsink);
  all_rates<A_8ACC_SGPR>(dcy, dsink);
  all_rates<A_ASHR64>(dcy, dsink);
  all_rates<A_MAD3_ADD1>(dcy, dsink);
  }
  std::vector<int32_t> hin(NL * 1024);
  srand(7);
  for (int i = 0; i < 1024; i++) {
    for (int j = 0; j < NL; j++) hin[NL * i + j] = (int32_t)((((uint32_t)rand() << 16) ^ rand()) & 0x0fffffff);
    hin[NL * i + NL - 1] = rand() & 0xffff;
  }
  CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  for (int wps = 1; wps <= 4; wps++) {
    run_mul<0, 1>(wps, dout, din, dcy, hin);
    run_mul<1, 1>(wps, dout, din, dcy, hin);
    run_mul<2, 1>(wps, dout, din, dcy, hin);
    run_mul<4, 1>(wps, dout, din, dcy, hin);
  }
  for (int wps = 5; wps <= 8; wps++) run_mul<4, 1>(wps, dout, din, dcy, hin);   // 60 VGPRs: up to eight waves per SIMD
  // the same with the register budget of four waves per SIMD forced
  for (int wps = 2; wps <= 4; wps += 2) {
    run_mul<0, 4>(wps, dout, din, dcy, hin);
    run_mul<1, 4>(wps, dout, din, dcy, hin);
    run_mul<2, 4>(wps, dout, din, dcy, hin);
    run_mul<4, 4>(wps, dout, din, dcy, hin);
  }
  return 0;
}
