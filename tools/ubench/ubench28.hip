// Microbenchmark for the carry-free Fp representation: 14 signed limbs of 28 bits, Montgomery R = 2^392, product scanning
// with ONE v_mad_i64_i32 per partial product (no carry-out folding, no SGPR carry hazards).  Measures Fp-multiplication
// throughput (1 stream) and the fused two-product pass (2 streams) at 1/2/4 waves per SIMD and checks the GPU limbs
// against the same algorithm on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench28 tools/ubench/ubench28.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define NL 14
#define MASK28 0x0fffffff
#define N0INV28 0xffcfffdu
#define HD __host__ __device__ __forceinline__

struct fp28 { int32_t l[NL]; };

__device__ __constant__ const int32_t P28_D[NL] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};
static const int32_t P28_H[NL] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void mad64(int64_t& acc, int32_t a, int32_t b) { uint64_t cy; asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mad64s(int64_t& acc, int32_t a, int32_t b) { uint64_t cy; asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "s"(b)); }
#else
static inline void mad64(int64_t& acc, int32_t a, int32_t b) { acc += (int64_t)a * b; }
static inline void mad64s(int64_t& acc, int32_t a, int32_t b) { acc += (int64_t)a * b; }
#endif
// r = REDC(a*b [+ c*d])
template <int STREAMS>
HD void fp28_mul(fp28& r, const fp28& a_, const fp28& b_, const fp28& c_, const fp28& d_, const int32_t* P) {
  fp28 a = a_, b = b_, c = c_, d = d_;
#if defined(__HIP_DEVICE_COMPILE__)
  // keep the operands opaque 32-bit values in this basic block so every product selects v_mad_i64_i32
#pragma unroll
  for (int i = 0; i < NL; i++) {
    asm volatile("" : "+v"(a.l[i]), "+v"(b.l[i]));
    if (STREAMS == 2) asm volatile("" : "+v"(c.l[i]), "+v"(d.l[i]));
  }
#endif
  int64_t acc = 0, acc2 = 0;
  int32_t m[NL];
  fp28 t;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    const int lo = k > NL - 1 ? k - (NL - 1) : 0, hi = k < NL - 1 ? k : NL - 1;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      acc += (int64_t)a.l[i] * b.l[k - i];
      if (STREAMS == 2) acc2 += (int64_t)c.l[i] * d.l[k - i];
    }
    if (STREAMS == 2) { acc += acc2; acc2 = 0; }
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (k < NL && i == k) continue;
      acc += (int64_t)m[i] * P[k - i];
    }
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
  r = t;
}

template <int STREAMS, int NCH>
__global__ void __launch_bounds__(256) k_mul(int32_t* out, const int32_t* in, int iters, uint64_t* cycles) {
  int id = blockIdx.x * blockDim.x + threadIdx.x;
  fp28 x[NCH], y, z;
  for (int c = 0; c < NCH; c++)
    for (int i = 0; i < NL; i++) x[c].l[i] = in[NL * ((id + c * 7) & 1023) + i];
  for (int i = 0; i < NL; i++) y.l[i] = in[NL * ((id + 3) & 1023) + i];
  for (int i = 0; i < NL; i++) z.l[i] = in[NL * ((id + 5) & 1023) + i];
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < NCH; c++) fp28_mul<STREAMS>(x[c], x[c], y, z, x[c], P28_D);
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  fp28 acc = x[0];
  for (int c = 1; c < NCH; c++)
    for (int i = 0; i < NL; i++) acc.l[i] += x[c].l[i];
  for (int i = 0; i < NL; i++) out[NL * id + i] = acc.l[i];
  if ((threadIdx.x & 63) == 0) cycles[id >> 6] = t1 - t0;
}

template <int STREAMS, int NCH>
static void run(int wps, int32_t* dout, int32_t* din, uint64_t* dcy, const std::vector<int32_t>& hin) {
  int blocks = 256 * wps, iters = 200;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_mul<STREAMS, NCH><<<blocks, 256>>>(dout, din, iters, dcy);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_mul<STREAMS, NCH><<<blocks, 256>>>(dout, din, iters, dcy);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<int32_t> o(NL * 1024);
  CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
  // host replay of the first 64 lanes
  bool ok = true;
  for (int id = 0; id < 64 && ok; id++) {
    fp28 x[NCH], y, z;
    for (int c = 0; c < NCH; c++) for (int i = 0; i < NL; i++) x[c].l[i] = hin[NL * ((id + c * 7) & 1023) + i];
    for (int i = 0; i < NL; i++) y.l[i] = hin[NL * ((id + 3) & 1023) + i];
    for (int i = 0; i < NL; i++) z.l[i] = hin[NL * ((id + 5) & 1023) + i];
    for (int it = 0; it < iters; it++) for (int c = 0; c < NCH; c++) fp28_mul<STREAMS>(x[c], x[c], y, z, x[c], P28_H);
    fp28 acc = x[0];
    for (int c = 1; c < NCH; c++) for (int i = 0; i < NL; i++) acc.l[i] += x[c].l[i];
    for (int i = 0; i < NL; i++) if (acc.l[i] != o[NL * id + i]) ok = false;
  }
  double nmul = (double)iters * NCH;
  printf("fp28 streams=%d chains=%d wps=%d  wall=%.3f ms  chip=%.2f G pass/s = %.2f G fp_mul-equiv/s  %s\n", STREAMS, NCH, wps, ms,
         (double)blocks * 256 * nmul / (ms * 1e-3) / 1e9, (double)blocks * 256 * nmul * (STREAMS == 2 ? 1.5 : 1.0) / (ms * 1e-3) / 1e9,
         ok ? "[matches host]" : "[MISMATCH vs host]");
}

int main() {
  uint64_t* dcy; int32_t *din, *dout;
  CK(hipMalloc(&dcy, 8 * 4 * 256 * 8));
  CK(hipMalloc(&din, NL * 4 * 1024)); CK(hipMalloc(&dout, NL * 4 * 256 * 256 * 8));
  std::vector<int32_t> hin(NL * 1024);
  srand(7);
  for (int i = 0; i < 1024; i++) {
    for (int j = 0; j < NL; j++) hin[NL * i + j] = (int32_t)((((uint32_t)rand() << 16) ^ rand()) & 0x1fffffff) - (1 << 27);  // lazy signed limbs
    hin[NL * i + NL - 1] = rand() & 0xffff;                                                                                     // value < p
  }
  CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  for (int wps = 1; wps <= 4; wps *= 2) {
    run<1, 1>(wps, dout, din, dcy, hin);
    run<1, 3>(wps, dout, din, dcy, hin);
    run<2, 1>(wps, dout, din, dcy, hin);
  }
  return 0;
}
