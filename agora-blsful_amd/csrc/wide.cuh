// Wide Fp: ONE field element spread over the 16 lanes of a DPP row -- lane j holds limb j (28-bit signed lazy limbs, the
// representation of fp.cuh; lanes 14 and 15 hold zero), so a wave carries four elements and an element costs one VGPR.
//
// Why: a single verification is a latency problem.  A lone wave issues one instruction every ~5 cycles whatever its
// type (profiles/ubench_r01_rates.txt), so the time of a dependent chain is its INSTRUCTION COUNT.  The lane-local
// multiplier (fp_mul_leaf) is ~500 instructions and every linear step around it (additions, carry passes, reductions,
// selections) costs 14 to 100 more; in the wave-cooperative pairing those linear steps are more than half of all
// instructions.  Across a row the same multiplication is ~225 instructions (14 lanes share the 392 multiply-adds) and an
// addition is ONE instruction, a carry pass five.
//
// Multiplication.  Columns T_k = sum_{i+j=k} a_i b_j: lane k accumulates T_k (k < 16) and T_{k+16} in two 64-bit
// accumulators; a_i is broadcast across the row (DPP row_newbcast), b shifted (row_shr / row_shl), one v_mad_i64_i32 per
// partial-product column pair.  Montgomery reduction (R = 2^392, as fp.cuh) word by word: step s broadcasts column s,
// every lane derives the same quotient digit q_s and adds q_s * p into its two columns; the running carry lives replicated
// on every lane, so no lane-selective operation is needed.  Two parallel carry passes leave the limbs normalised.
// Device only (DPP); checked against the oracle through blsgpu_debug_wide_mul (tests/test_gpu_wide.py).
#pragma once
#include "fp.cuh"

typedef int32_t wfp;     // this lane's limb of a wide element
#define FP_P0_CONST 0x0fffaaab   // limb 0 of the modulus (FP_P[0]) as an immediate

template <int N>
__device__ __forceinline__ int32_t w_shr(int32_t v) {   // lane k <- lane k - N of the row (0 below the row start)
  if (N == 0) return v;
  return __builtin_amdgcn_update_dpp(0, v, 0x110 + (N ? N : 1), 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ int32_t w_shl(int32_t v) {   // lane k <- lane k + N of the row (0 beyond the row end)
  if (N == 0) return v;
  return __builtin_amdgcn_update_dpp(0, v, 0x100 + (N ? N : 1), 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ int32_t w_bcast(int32_t v) {  // every lane of the row <- lane N of the row (gfx90a+ row_newbcast)
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, true);
}

// per-lane constants of the reduction: the modulus shifted to every step's position
struct wide_consts {
  int32_t plo[FP_NL], phi[FP_NL];   // lane k: p_{k-s} and p_{k+16-s} (0 outside 0..13)
  int32_t keep;                      // limb mask of a carry pass: 2^28 - 1 on lanes 0..12, all ones on lane 13 (the signed top limb), 0 above
  int32_t pass;                      // all ones on lanes 0..12: their carries move up; 0 from lane 13 on
  int lane;                          // lane index inside the row
};
__device__ __forceinline__ void wide_init(wide_consts& K) {
  const int l = (int)(threadIdx.x & 15u);
  K.lane = l;
#pragma unroll
  for (int s = 0; s < FP_NL; s++) {
    const int a = l - s, b = l + 16 - s;
    K.plo[s] = (a >= 0 && a < FP_NL) ? (int32_t)FP_P[a] : 0;
    K.phi[s] = (b >= 0 && b < FP_NL) ? (int32_t)FP_P[b] : 0;
  }
  K.keep = l < FP_NL - 1 ? (int32_t)FP_MASK : (l == FP_NL - 1 ? -1 : 0);
  K.pass = l < FP_NL - 1 ? -1 : 0;
}

// one parallel carry pass (value unchanged): limbs with magnitude below 2^31 come back below 2^28 + 8
__device__ __forceinline__ wfp w_norm(wfp v, const wide_consts& K) {
  const int32_t c = (v >> FP_LB) & K.pass;
  return (v & K.keep) + w_shr<1>(c);
}

template <int S>
__device__ __forceinline__ void w_mul_prod_step(int64_t& lo, int64_t& hi, wfp a, wfp b) {
  const int32_t ai = w_bcast<S>(a);
  lo += (int64_t)ai * w_shr<S>(b);
  if (S >= 3) hi += (int64_t)ai * w_shl<(S >= 3 ? 16 - S : 1)>(b);
}
template <int S>
__device__ __forceinline__ void w_mul_redc_step(int64_t& lo, int64_t& hi, int64_t& cy, const wide_consts& K) {
  const uint32_t l0 = (uint32_t)w_bcast<S>((int32_t)(uint32_t)lo);
  const int32_t l1 = w_bcast<S>((int32_t)(lo >> 32));
  const int64_t col = (int64_t)(((uint64_t)(uint32_t)l1 << 32) | l0) + cy;
  const int32_t q = (int32_t)(((uint32_t)col * FP_N0INV) & FP_MASK);
  lo += (int64_t)q * K.plo[S];
  hi += (int64_t)q * K.phi[S];
  cy = (col + (int64_t)q * (int32_t)FP_P0_CONST) >> FP_LB;
}
// a * b / R mod p; operand limbs: |a_i| |b_j| summed over a column must stay below 2^62 (e.g. both below 2^29.5)
__device__ __forceinline__ wfp w_mul(wfp a, wfp b, const wide_consts& K) {
  int64_t lo = 0, hi = 0, cy = 0;
  w_mul_prod_step<0>(lo, hi, a, b);
  w_mul_prod_step<1>(lo, hi, a, b);
  w_mul_prod_step<2>(lo, hi, a, b);
  w_mul_prod_step<3>(lo, hi, a, b);
  w_mul_prod_step<4>(lo, hi, a, b);
  w_mul_prod_step<5>(lo, hi, a, b);
  w_mul_prod_step<6>(lo, hi, a, b);
  w_mul_prod_step<7>(lo, hi, a, b);
  w_mul_prod_step<8>(lo, hi, a, b);
  w_mul_prod_step<9>(lo, hi, a, b);
  w_mul_prod_step<10>(lo, hi, a, b);
  w_mul_prod_step<11>(lo, hi, a, b);
  w_mul_prod_step<12>(lo, hi, a, b);
  w_mul_prod_step<13>(lo, hi, a, b);
  w_mul_redc_step<0>(lo, hi, cy, K);
  w_mul_redc_step<1>(lo, hi, cy, K);
  w_mul_redc_step<2>(lo, hi, cy, K);
  w_mul_redc_step<3>(lo, hi, cy, K);
  w_mul_redc_step<4>(lo, hi, cy, K);
  w_mul_redc_step<5>(lo, hi, cy, K);
  w_mul_redc_step<6>(lo, hi, cy, K);
  w_mul_redc_step<7>(lo, hi, cy, K);
  w_mul_redc_step<8>(lo, hi, cy, K);
  w_mul_redc_step<9>(lo, hi, cy, K);
  w_mul_redc_step<10>(lo, hi, cy, K);
  w_mul_redc_step<11>(lo, hi, cy, K);
  w_mul_redc_step<12>(lo, hi, cy, K);
  w_mul_redc_step<13>(lo, hi, cy, K);
  // result column 14 + j -> lane j: columns 14, 15 sit in lo of lanes 14, 15; column 16 + k in hi of lane k
  const uint32_t r0 = (uint32_t)w_shl<14>((int32_t)(uint32_t)lo) | (uint32_t)w_shr<2>((int32_t)(uint32_t)hi);
  const int32_t r1 = w_shl<14>((int32_t)(lo >> 32)) | w_shr<2>((int32_t)(hi >> 32));
  int64_t v = (int64_t)(((uint64_t)(uint32_t)r1 << 32) | r0);
  if (K.lane == 0) v += cy;
  // first carry pass on 64-bit columns (below 2^62), second on 32 bits
  const int64_t c1 = v >> FP_LB;
  const int32_t c1lo = (int32_t)(uint32_t)c1 & K.pass, c1hi = (int32_t)(c1 >> 32) & K.pass;
  const int64_t in1 = (int64_t)(((uint64_t)(uint32_t)w_shr<1>(c1hi) << 32) | (uint32_t)w_shr<1>(c1lo));
  const int64_t keep64 = K.lane < FP_NL - 1 ? (int64_t)FP_MASK : (K.lane == FP_NL - 1 ? (int64_t)-1 : (int64_t)0);
  const int64_t v1 = (v & keep64) + in1;                 // below 2^28 + 2^34 on lanes 0..12, the small signed top limb on lane 13
  const int32_t c2 = (int32_t)(v1 >> FP_LB) & K.pass;
  return ((int32_t)v1 & K.keep) + w_shr<1>(c2);
}
__device__ __forceinline__ wfp w_sqr(wfp a, const wide_consts& K) { return w_mul(a, a, K); }

// lane-local fp <-> wide through LDS: `slot` points at 16 words owned by the row
__device__ __forceinline__ void w_store_local(uint32_t* slot, const fp& a) {   // one lane writes all 14 limbs (and the two zeros)
#pragma unroll
  for (int i = 0; i < FP_NL; i++) slot[i] = (uint32_t)a.l[i];
  slot[14] = 0;
  slot[15] = 0;
}
__device__ __forceinline__ void w_load_local(fp& r, const uint32_t* slot) {
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = (int32_t)slot[i];
}
