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
// ---- the two halves of a multiplication on their own, for sums of products that are reduced ONCE (wide_engine.cuh):
// w_mul_cols: the 27 columns of a * b, not reduced, carried once so that every column fits a signed 32-bit word (below
// 2^29.6): lane k holds column k in clo and column 16 + k in chi.  ~80 instructions.
__device__ __forceinline__ void w_mul_cols(int32_t& clo, int32_t& chi, wfp a, wfp b, const wide_consts& K) {
  int64_t lo = 0, hi = 0;
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
  // a column c (below 2^63 in magnitude) = f0 + f1 2^28 + f2 2^56 with f0, f1 28-bit fields and f2 the signed rest: f1 moves
  // one column up, f2 two.  Column 26 (chi of lane 10, the single product a13 b13, below 2^36) hands its whole signed rest to
  // column 27; nothing sits above it.
  const int32_t l0 = (int32_t)(uint32_t)lo, l1 = (int32_t)(lo >> 32), h0 = (int32_t)(uint32_t)hi, h1 = (int32_t)(hi >> 32);
  const int32_t lf0 = l0 & (int32_t)FP_MASK, lf1 = (int32_t)__builtin_amdgcn_alignbit((uint32_t)l1, (uint32_t)l0, FP_LB) & (int32_t)FP_MASK, lf2 = l1 >> 24;
  const int32_t hr = (int32_t)__builtin_amdgcn_alignbit((uint32_t)h1, (uint32_t)h0, FP_LB);
  const int32_t hf0 = h0 & (int32_t)FP_MASK, hf1 = K.lane < 10 ? (hr & (int32_t)FP_MASK) : (K.lane == 10 ? hr : 0), hf2 = K.lane < 10 ? (h1 >> 24) : 0;
  clo = lf0 + w_shr<1>(lf1) + w_shr<2>(lf2);
  chi = hf0 + w_shr<1>(hf1) + w_shr<2>(hf2) + w_shl<15>(lf1) + w_shl<14>(lf2);   // columns 15 -> 16, 14 -> 16, 15 -> 17
}
// w_redc_cols: Montgomery reduction of 28 columns (lane k: column k in lo, column 16 + k in hi; magnitudes below 2^62) to
// the 64-bit limb sums of T / R mod p (lane j: limb j, below 2^40), ready for w_finish / a carry pass
__device__ __forceinline__ int64_t w_redc_cols(int64_t lo, int64_t hi, const wide_consts& K) {
  int64_t cy = 0;
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
  const uint32_t r0 = (uint32_t)w_shl<14>((int32_t)(uint32_t)lo) | (uint32_t)w_shr<2>((int32_t)(uint32_t)hi);
  const int32_t r1 = w_shl<14>((int32_t)(lo >> 32)) | w_shr<2>((int32_t)(hi >> 32));
  int64_t v = (int64_t)(((uint64_t)(uint32_t)r1 << 32) | r0);
  if (K.lane == 0) v += cy;
  return v;
}


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

// ---------------------------------------------------------------------------------------------------------------------
// `wf`: the wide element as a field type with the interface of `fp` (fp_mul, fp_add, ..., the fe_* names of curve.cuh), so
// that the templated curve code (jac<F>: additions, doublings, the cofactor multiplication) and the SSWU map run on rows
// unchanged: every row of a wave is an independent instance, all rows execute the same instruction stream.  Predicates
// that need the canonical form (is-zero, parity) send the element through LDS to one lane's lane-local code: they are
// rare (a few dozen per hash), multiplications are not.
// Per-workgroup LDS of the wide field layer (WIDE_ROWS rows): the reduction's per-lane constants, a scratch slot per row,
// the window table of an exponentiation per row.
#ifndef WIDE_ROWS
#define WIDE_ROWS 16
#endif
struct wide_field_lds {
  int32_t k_plo[FP_NL][16], k_phi[FP_NL][16];
  uint32_t slot[WIDE_ROWS][16];
  int32_t pred[WIDE_ROWS];
  uint32_t tab[WIDE_ROWS][16][16];
};
__shared__ wide_field_lds g_wf;

struct wf {
  wfp v;
};
__device__ __forceinline__ int wf_lane() { return (int)(threadIdx.x & 15u); }
__device__ __forceinline__ int wf_row() { return (int)(threadIdx.x >> 4); }
// call once per kernel from all threads (then a barrier) before any wf arithmetic
__device__ __forceinline__ void wf_setup() {
  for (unsigned t = threadIdx.x; t < FP_NL * 16; t += blockDim.x) {
    const int s = t >> 4, l = t & 15, a = l - s, b = l + 16 - s;
    g_wf.k_plo[s][l] = (a >= 0 && a < FP_NL) ? (int32_t)FP_P[a] : 0;
    g_wf.k_phi[s][l] = (b >= 0 && b < FP_NL) ? (int32_t)FP_P[b] : 0;
  }
}
__device__ __forceinline__ void wf_consts(wide_consts& K) {
  const int l = wf_lane();
  K.lane = l;
#pragma unroll
  for (int s = 0; s < FP_NL; s++) {
    K.plo[s] = g_wf.k_plo[s][l];
    K.phi[s] = g_wf.k_phi[s][l];
  }
  K.keep = l < FP_NL - 1 ? (int32_t)FP_MASK : (l == FP_NL - 1 ? -1 : 0);
  K.pass = l < FP_NL - 1 ? -1 : 0;
}
// the one out-of-line multiplier of the wide field layer (~270 instructions with the constant loads)
__device__ __noinline__ wfp wf_mul_leaf(wfp a, wfp b) {
  wide_consts K;
  wf_consts(K);
  return w_mul(a, b, K);
}
__device__ __forceinline__ int32_t wf_keep() { const int l = wf_lane(); return l < FP_NL - 1 ? (int32_t)FP_MASK : (l == FP_NL - 1 ? -1 : 0); }
__device__ __forceinline__ int32_t wf_pass() { return wf_lane() < FP_NL - 1 ? -1 : 0; }
__device__ __forceinline__ wfp wf_norm1(wfp v) {
  const int32_t c = (v >> FP_LB) & wf_pass();
  return (v & wf_keep()) + w_shr<1>(c);
}
__device__ __forceinline__ void fp_zero(wf& r) { r.v = 0; }
__device__ __forceinline__ void fp_load(wf& r, const uint32_t* c) { const int l = wf_lane(); r.v = l < FP_NL ? (int32_t)c[l] : 0; }
__device__ __forceinline__ void fp_one(wf& r) { fp_load(r, FP_ONE); }
__device__ __forceinline__ void fp_add(wf& r, const wf& a, const wf& b) { r.v = a.v + b.v; }
__device__ __forceinline__ void fp_sub(wf& r, const wf& a, const wf& b) { r.v = a.v - b.v; }
__device__ __forceinline__ void fp_neg(wf& r, const wf& a) { r.v = -a.v; }
__device__ __forceinline__ void fp_dbl(wf& r, const wf& a) { r.v = a.v + a.v; }
__device__ __forceinline__ void fp_norm(wf& r, const wf& a) { r.v = wf_norm1(a.v); }
__device__ __forceinline__ void fp_cmov(wf& r, const wf& a, bool c) { r.v = c ? a.v : r.v; }
// value brought back to (-0.6 p, 0.6 p) (nearest multiple of p estimated from the top limb), limbs normalised; the input may
// carry limbs up to 2^31 in magnitude and a value of a few dozen p
__device__ __forceinline__ void fp_reduce(wf& r, const wf& a) {
  const wfp v1 = wf_norm1(a.v);
  const int32_t top = w_bcast<13>(v1);
  const int32_t k = __float2int_rn((float)top * (1.0f / 106513.57f));
  const int l = wf_lane();
  const int32_t pl = l < FP_NL ? (int32_t)FP_P[l] : 0;
  const int64_t t = (int64_t)v1 - (int64_t)k * pl;
  const int32_t c2 = (int32_t)(t >> FP_LB) & wf_pass();
  const int64_t keep64 = l < FP_NL - 1 ? (int64_t)FP_MASK : (l == FP_NL - 1 ? (int64_t)-1 : (int64_t)0);
  r.v = wf_norm1((int32_t)(t & keep64) + w_shr<1>(c2));
}
// operands of a product must have limbs below ~2^29.5: sums of up to two normalised values qualify, longer sums are
// normalised first (one carry pass)
__device__ __forceinline__ void fp_mul(wf& r, const wf& a, const wf& b) { r.v = wf_mul_leaf(a.v, b.v); }
__device__ __forceinline__ void fp_sqr(wf& r, const wf& a) { r.v = wf_mul_leaf(a.v, a.v); }

// element of this row -> lane-local fp on every lane of the row (through the row's LDS slot)
__device__ __forceinline__ void wf_to_local(fp& r, const wf& a) {
  uint32_t* s = g_wf.slot[wf_row()];
  s[wf_lane()] = (uint32_t)a.v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  w_load_local(r, s);
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wf_from_local(wf& r, const fp& a) {   // every lane of the row holds the same lane-local value
  const int l = wf_lane();
  int32_t v = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) v = (l == i) ? a.l[i] : v;
  r.v = v;
}
__device__ __noinline__ bool wf_is_zero_leaf(wfp v) {
  fp t;
  wf a;
  a.v = v;
  wf_to_local(t, a);
  return fp_is_zero(t);
}
__device__ __noinline__ uint32_t wf_parity_leaf(wfp v) {
  fp t;
  wf a;
  a.v = v;
  wf_to_local(t, a);
  return fp_parity(t);
}
__device__ __forceinline__ bool fp_is_zero(const wf& a) { return wf_is_zero_leaf(a.v); }
__device__ __forceinline__ bool fp_eq(const wf& a, const wf& b) { return wf_is_zero_leaf(a.v - b.v); }
__device__ __forceinline__ uint32_t fp_parity(const wf& a) { return wf_parity_leaf(a.v); }

// a^e for a public exponent: sliding 5-bit windows over the sixteen odd powers (fp.cuh fp_pow; 375 + 66 + 16 operations for
// (p-3)/4); the table lives in the row's LDS block, the accumulator in a register
__device__ __noinline__ void fp_pow(wf& r, const wf& a, const uint32_t* e, int nbits) {
  uint32_t (*tab)[16] = g_wf.tab[wf_row()];
  const int l = wf_lane();
  wide_consts K;
  wf_consts(K);
  wf a1;
  fp_norm(a1, a);
  const wfp a2 = w_mul(a1.v, a1.v, K);
  wfp t = a1.v;
  tab[0][l] = (uint32_t)t;
  for (int i = 1; i < 16; i++) {
    t = w_mul(t, a2, K);
    tab[i][l] = (uint32_t)t;
  }
  wfp acc = 0;
  bool started = false;
  pow_bits x = {e, -1, 0};
  for (int i = nbits - 1; i >= 0;) {
    pow_seek(x, i);
    if (!pow_bit(x, i)) {
      if (started) acc = w_mul(acc, acc, K);
      i--;
      continue;
    }
    uint32_t v;
    const int j = pow_window(x, i, 5, v);
    if (started) {
      for (int k = i; k >= j; k--) acc = w_mul(acc, acc, K);
      acc = w_mul(acc, (wfp)tab[v >> 1][l], K);
    } else {
      acc = (wfp)tab[v >> 1][l];
      started = true;
    }
    i = j - 1;
  }
  if (!started) {
    wf one;
    fp_one(one);
    acc = one.v;
  }
  r.v = acc;
}

// the fe_* names of curve.cuh
__device__ __forceinline__ void fe_add(wf& r, const wf& a, const wf& b) { fp_add(r, a, b); }
__device__ __forceinline__ void fe_sub(wf& r, const wf& a, const wf& b) { fp_sub(r, a, b); }
__device__ __forceinline__ void fe_mul(wf& r, const wf& a, const wf& b) { fp_mul(r, a, b); }
__device__ __forceinline__ void fe_sqr(wf& r, const wf& a) { fp_sqr(r, a); }
__device__ __forceinline__ void fe_neg(wf& r, const wf& a) { fp_neg(r, a); }
__device__ __forceinline__ void fe_dbl(wf& r, const wf& a) { fp_dbl(r, a); }
__device__ __forceinline__ bool fe_is_zero(const wf& a) { return fp_is_zero(a); }
__device__ __forceinline__ bool fe_eq(const wf& a, const wf& b) { return fp_eq(a, b); }
__device__ __forceinline__ void fe_zero(wf& r) { fp_zero(r); }
__device__ __forceinline__ void fe_one(wf& r) { fp_one(r); }
__device__ __forceinline__ void fe_cmov(wf& r, const wf& a, bool c) { fp_cmov(r, a, c); }
__device__ __forceinline__ void fe_norm(wf& r, const wf& a) { fp_norm(r, a); }
__device__ __forceinline__ void fe_reduce(wf& r, const wf& a) { fp_reduce(r, a); }
