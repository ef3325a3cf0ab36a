// Round-4 microbenchmark (optional): can the matrix cores carry the CONSTANT half of a Montgomery multiplication?
// The reduction's m * p (m: the fourteen 28-bit quotient digits of a lane's item, p: the modulus) is a product with a constant Toeplitz
// matrix.  Two ways to get the 28 limbs of m * p for the 64 items of a wave (one item per lane, as every lane-local kernel of the library holds
// its data):
//   valu   196 v_mad_i64_i32 (product scanning, one 64-bit accumulator) + the carry chain: what fp_redc_products spends on m * p.
//   mfma   v_mfma_i32_16x16x64_i8 on 7-bit digits (a 28-bit limb is exactly four of them; i8 operands are signed, so bytes do not fit):
//          pack 14 limbs into 56 digits (4 per dword), transpose through LDS into the B-operand layout (16 items per instruction: lane (q, j) holds
//          digits 16 q .. 16 q + 15 of item j of the group), 4 groups x 7 column blocks = 28 instructions against the constant digit-Toeplitz
//          blocks of p held in 28 registers, fold the four 7-bit columns a lane holds per block into one lazy 64-bit limb (limb 4 b + q), transpose
//          back through LDS, one carry chain.
// Both leave the same 28 exact limbs (checked against the host); the next iteration's m is derived from them, so iterations depend on each other.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ubench5 tools/ubench/ubench5.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define NL 14
#define MASK28 0x0fffffffu
typedef int v4i __attribute__((ext_vector_type(4)));
static const uint32_t P28_H[NL] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};
__device__ __constant__ const uint32_t P28_D[NL] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};

// the 28 limbs of m * p (exact), the next m = low half XOR high half
__host__ __device__ inline void next_m(uint32_t* m, const uint32_t* n) {
  for (int i = 0; i < NL; i++) m[i] = (n[i] ^ n[NL + i]) & MASK28;
}
static void host_iter(uint32_t* m) {
  uint32_t n[2 * NL];
  uint64_t acc = 0;
  for (int k = 0; k < 2 * NL - 1; k++) {
    const int lo = k > NL - 1 ? k - (NL - 1) : 0, hi = k < NL - 1 ? k : NL - 1;
    for (int i = lo; i <= hi; i++) acc += (uint64_t)m[i] * P28_H[k - i];
    n[k] = (uint32_t)acc & MASK28;
    acc >>= 28;
  }
  n[2 * NL - 1] = (uint32_t)acc;
  next_m(m, n);
}

template <int LB>
__global__ void __launch_bounds__(64, LB) k_valu(uint32_t* out, const uint32_t* in, int iters, uint64_t* cycles) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m[NL];
  for (int i = 0; i < NL; i++) m[i] = in[NL * (id & 1023) + i];
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    uint32_t n[2 * NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
      const int lo = k > NL - 1 ? k - (NL - 1) : 0, hi = k < NL - 1 ? k : NL - 1;
#pragma unroll
      for (int i = lo; i <= hi; i++) acc += (uint64_t)m[i] * P28_D[k - i];
      n[k] = (uint32_t)acc & MASK28;
      acc >>= 28;
    }
    n[2 * NL - 1] = (uint32_t)acc;
    next_m(m, n);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < NL; i++) out[NL * id + i] = m[i];
  if ((threadIdx.x & 63) == 0) cycles[id >> 6] = t1 - t0;
}

// 7-bit digits of the modulus: digit k of p (k < 55), 0 outside
__device__ __forceinline__ uint32_t p_digit(int k) {
  if (k < 0 || k >= 4 * NL) return 0;
  return (P28_D[k >> 2] >> (7 * (k & 3))) & 0x7fu;
}
__device__ __forceinline__ uint32_t pack7(uint32_t l) {      // four 7-bit digits of a 28-bit limb, one per byte
  return (l & 0x7fu) | ((l & 0x3f80u) << 1) | ((l & 0x1fc000u) << 2) | ((l & 0xfe00000u) << 3);
}
#define IN_STRIDE 16      // dwords per item in the operand image
#define OUT_STRIDE 29     // 64-bit words per item in the result image (odd: the 64-bit columns of consecutive items fall on different banks)
template <int LB>
__global__ void __launch_bounds__(64, LB) k_mfma(uint32_t* out, const uint32_t* in, int iters, uint64_t* cycles) {
  __shared__ __attribute__((aligned(16))) uint32_t sin_[64 * IN_STRIDE];
  __shared__ uint64_t sout[64 * OUT_STRIDE];
  const int id = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x, q = lane >> 4, j = lane & 15;
  uint32_t m[NL];
  for (int i = 0; i < NL; i++) m[i] = in[NL * (id & 1023) + i];
  // this lane's fragments of the seven constant blocks: A_b[row i = j][k = 16 q + e] = digit (16 b + i - k) of p, byte e of the fragment
  v4i A[7];
  for (int b = 0; b < 7; b++) {
    uint32_t w[4];
    for (int r = 0; r < 4; r++) {
      uint32_t x = 0;
      for (int e = 0; e < 4; e++) x |= p_digit(16 * b + j - (16 * q + 4 * r + e)) << (8 * e);
      w[r] = x;
    }
    A[b] = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  }
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    // (1) digits, (2) this lane's row of the operand image
    uint32_t d[16];
#pragma unroll
    for (int t = 0; t < NL; t++) d[t] = pack7(m[t]);
    d[14] = 0;
    d[15] = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) *(uint4*)&sin_[lane * IN_STRIDE + 4 * r] = make_uint4(d[4 * r], d[4 * r + 1], d[4 * r + 2], d[4 * r + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (3) per group of sixteen items: the B fragment (digits 16 q .. 16 q + 15 of item j of the group), seven blocks of sixteen columns
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const uint4 bw = *(const uint4*)&sin_[(16 * g + j) * IN_STRIDE + 4 * q];
      const v4i B = v4i{(int)bw.x, (int)bw.y, (int)bw.z, (int)bw.w};
#pragma unroll
      for (int b = 0; b < 7; b++) {
        const v4i z = {0, 0, 0, 0};
        const v4i D = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[b], B, z, 0, 0, 0);
        // rows 4 q + r of block b: the four 7-bit columns of limb 4 b + q of item (g, j): one lazy 64-bit limb
        const uint64_t L = (uint64_t)(uint32_t)D[0] + ((uint64_t)(uint32_t)D[1] << 7) + ((uint64_t)(uint32_t)D[2] << 14) + ((uint64_t)(uint32_t)D[3] << 21);
        sout[(16 * g + j) * OUT_STRIDE + 4 * b + q] = L;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (4) this lane's item: 28 lazy limbs, one carry chain
    uint32_t n[2 * NL];
    uint64_t c = 0;
#pragma unroll
    for (int t = 0; t < 2 * NL; t++) {
      c += sout[lane * OUT_STRIDE + t];
      n[t] = t < 2 * NL - 1 ? ((uint32_t)c & MASK28) : (uint32_t)c;
      c >>= 28;
    }
    __builtin_amdgcn_wave_barrier();
    next_m(m, n);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < NL; i++) out[NL * id + i] = m[i];
  if ((threadIdx.x & 63) == 0) cycles[id >> 6] = t1 - t0;
}

template <class K>
static void run(const char* name, K kern, const void* fptr, int wps, uint32_t* dout, uint32_t* din, uint64_t* dcy, const std::vector<uint32_t>& want, int iters) {
  const int blocks = 1024 * wps;          // 64-thread workgroups: 1,024 SIMDs
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, 64>>>(dout, din, iters, dcy);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<blocks, 64>>>(dout, din, iters, dcy);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint32_t> o((size_t)NL * 1024);
  CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint64_t> cy(blocks);
  CK(hipMemcpy(cy.data(), dcy, cy.size() * 8, hipMemcpyDeviceToHost));
  std::sort(cy.begin(), cy.end());
  const bool ok = o == want;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, fptr));
  const double prods = (double)blocks * 64 * iters;
  printf("%-6s wps=%d  regs=%3d scratch=%3zu lds=%5zu  wall=%.3f ms  %.2f G (m x p)/s whole chip  wave cycles/iter=%.0f  %s\n", name, wps, fa.numRegs, (size_t)fa.localSizeBytes,
         (size_t)fa.sharedSizeBytes, ms, prods / (ms * 1e-3) / 1e9, (double)cy[cy.size() / 2] / iters, ok ? "[matches host]" : "[MISMATCH vs host]");
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  uint64_t* dcy; uint32_t *din, *dout;
  CK(hipMalloc(&dcy, 8 * 1024 * 8));
  CK(hipMalloc(&din, NL * 4 * 1024));
  CK(hipMalloc(&dout, (size_t)NL * 4 * 64 * 1024 * 8));
  std::vector<uint32_t> hin(NL * 1024);
  srand(11);
  for (auto& x : hin) x = (((uint32_t)rand() << 16) ^ (uint32_t)rand()) & MASK28;
  CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  const int iters = 200;
  std::vector<uint32_t> want = hin;
  for (int i = 0; i < 1024; i++)
    for (int it = 0; it < iters; it++) host_iter(&want[NL * i]);
  for (int wps = 1; wps <= 4; wps++) {
    if (wps <= 2) {
      run("valu", k_valu<2>, (const void*)k_valu<2>, wps, dout, din, dcy, want, iters);
      run("mfma", k_mfma<2>, (const void*)k_mfma<2>, wps, dout, din, dcy, want, iters);
    } else {
      run("valu", k_valu<4>, (const void*)k_valu<4>, wps, dout, din, dcy, want, iters);
      run("mfma", k_mfma<4>, (const void*)k_mfma<4>, wps, dout, din, dcy, want, iters);
    }
  }
  return 0;
}
