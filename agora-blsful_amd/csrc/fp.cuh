// Fp arithmetic for BLS12-381 on gfx950: fourteen SIGNED 28-bit limbs (one per 32-bit register), Montgomery form with
// R = 2^392, lazy (carry-free) additions.
//
// Why this shape (measured on MI355X, profiles/ubench_r01.txt + tools/ubench): on gfx950 every VOP3-encoded integer
// instruction -- v_mad_u64_u32, v_mul_lo_u32, and also v_addc_co_u32 with an SGPR carry -- costs the same issue slot
// (~4.5 cycles per wave-instruction at two waves per SIMD), a carry chain needs two wait states between links (hipcc pads
// every link of a 12-limb add with s_nop 1), and only VOP2 adds without carry run at ~2.4 cycles.  A saturated 12 x 32-bit
// representation therefore pays one multiply-class slot per partial product for the carry fold and ~250 cycles per
// modular addition.  With 28-bit limbs a column of 14 (or 28) products fits a 64-bit accumulator, so a product costs ONE
// v_mad_i64_i32 and nothing else; additions, subtractions and negations are 14 plain v_add/v_sub_u32 with no carries, no
// conditional subtraction and no hazards.  Signed limbs make subtraction as cheap as addition.
//
// Value discipline.  A limb vector (l_0..l_13) stands for the integer sum l_i 2^(28 i); an element is that integer mod p.
//   * fp_mul / fp_dotp2 return limbs 0..12 in [0, 2^28) and a small signed top limb; the value lies in (-p/8, p + p/8)
//     whenever the products of the operands' magnitudes sum to less than 256 p^2 (p / R = 2^-11.3).
//   * fp_add / fp_sub / fp_neg / fp_dbl are limb-wise and exact on the integers: bounds add up.
//   * A multiplication needs 14 * sum |a_i| |b_j| + 2^60 < 2^63 per column: operand limb bounds with
//     A * B <= 2^59 (one product stream) or 2^58 (two streams).  fp_norm (one parallel carry pass, 3 VOP2 per limb)
//     brings limbs back to 2^28 + a few units when a chain of additions gets too long (the value is unchanged).
//   * Additions also let the VALUE grow (nothing is subtracted), and a squaring chain doubles it every step, so the
//     tower functions (Fp6/Fp12 products, cyclotomic squaring, curve formulas) fp_reduce their results: one exact
//     carry pass that subtracts the nearest multiple of p -> limbs in [0, 2^28), value in (-0.52 p, 0.52 p).
//   * Predicates and serialisation (fp_is_zero, fp_eq, sgn0, compress, ...) go through fp_canon, the unique
//     representative in [0, p) with exact limbs.
// tests/hostsim compiles this header on the host with -DBLS_TRACK_BOUNDS: every fp then carries worst-case bounds
// (limb magnitude, value in units of p) that are propagated through every operation and checked against the rules above,
// so one pass over the verification paths proves the placement of the fp_norm calls for all inputs.
//
// The caller-facing RAW formats stay blst's 6 x u64 little-endian Montgomery words (R = 2^384):
// fp_from_raw / fp_to_raw convert with one multiplication (by 2^400 resp. 2^384 mod p) at the boundary.
//
// This header also compiles as plain C++ (no HIP) for the host-side unit tests in tests/hostsim.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BLS_FN __device__ __forceinline__
#define BLS_MFN __device__ __forceinline__   // member functions
#define BLS_NOINLINE __device__ __noinline__
#define BLS_CONST __device__ __constant__ const
#else
#define BLS_FN static inline
#define BLS_MFN inline
#define BLS_NOINLINE static __attribute__((noinline))
#define BLS_CONST static const
#endif

#include "consts.cuh"

#if defined(BLS_TRACK_BOUNDS) && !defined(__HIPCC__)
#include <stdio.h>
#include <stdlib.h>
#define FP_TRK(...) __VA_ARGS__
#define FP_LB_N 268435456.0   // 2^28: limb bound of a normalised element
static void fp_trk_fail(const char* what, double x, double y) {
  fprintf(stderr, "BLS_TRACK_BOUNDS: %s violated (%.4g, %.4g)\n", what, x, y);
  abort();
}
#else
#define FP_TRK(...)
#endif

struct fp {
  int32_t l[FP_NL];
  FP_TRK(double lb; double vb;      // worst-case |limb| and |value| / p
         bool nn = false;)         // limbs 0..12 known to lie in [0, 2^28) and the top limb within +-2^20 (a REDC / fp_reduce output): fp_lin4 bounds such operands by sign
};

// ---- basic ops -----------------------------------------------------------------------------------
// internal 14-word form (constants, device workspaces): limbs are normalised, value in (-p/8, p + p/8)
BLS_FN void fp_load(fp& r, const uint32_t* c) {
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = (int32_t)c[i];
  FP_TRK(r.lb = FP_LB_N; r.vb = 1.125; r.nn = false;)
}

BLS_FN void fp_store(uint32_t* c, const fp& a) {
  FP_TRK(if (a.lb > FP_LB_N || a.vb > 1.125) fp_trk_fail("fp_store of a normalised product", a.lb, a.vb);)
#pragma unroll
  for (int i = 0; i < FP_NL; i++) c[i] = (uint32_t)a.l[i];
}

BLS_FN void fp_zero(fp& r) {
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = 0;
  FP_TRK(r.lb = 0; r.vb = 0; r.nn = true;)
}

BLS_FN void fp_one(fp& r) { fp_load(r, FP_ONE); }

BLS_FN void fp_cmov(fp& r, const fp& a, bool c) {  // r = c ? a : r
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = c ? a.l[i] : r.l[i];
  FP_TRK(r.lb = fmax(r.lb, a.lb); r.vb = fmax(r.vb, a.vb); r.nn = r.nn && a.nn;)
}

BLS_FN void fp_add(fp& r, const fp& a, const fp& b) {
  FP_TRK(const double lb = a.lb + b.lb, vb = a.vb + b.vb; if (lb >= 2147483648.0) fp_trk_fail("fp_add limb < 2^31", a.lb, b.lb);)
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = a.l[i] + b.l[i];
  FP_TRK(r.lb = lb; r.vb = vb; r.nn = false;)
}

BLS_FN void fp_sub(fp& r, const fp& a, const fp& b) {
  FP_TRK(const double lb = a.lb + b.lb, vb = a.vb + b.vb; if (lb >= 2147483648.0) fp_trk_fail("fp_sub limb < 2^31", a.lb, b.lb);)
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = a.l[i] - b.l[i];
  FP_TRK(r.lb = lb; r.vb = vb; r.nn = false;)
}

BLS_FN void fp_neg(fp& r, const fp& a) {
  FP_TRK(const double lb = a.lb, vb = a.vb;)
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = -a.l[i];
  FP_TRK(r.lb = lb; r.vb = vb; r.nn = false;)
}

BLS_FN void fp_dbl(fp& r, const fp& a) { fp_add(r, a, a); }

// one parallel carry pass: limbs 0..12 -> [-2^(B-28), 2^28 + 2^(B-28)), the value is unchanged
BLS_FN void fp_norm(fp& r, const fp& a) {
  FP_TRK(const double lb = FP_LB_N + floor(a.lb / FP_LB_N) + 1, vb = a.vb;)
  int32_t c[FP_NL - 1];
#pragma unroll
  for (int i = 0; i < FP_NL - 1; i++) c[i] = a.l[i] >> FP_LB;
  const int32_t top = a.l[FP_NL - 1] + c[FP_NL - 2];
#pragma unroll
  for (int i = FP_NL - 2; i >= 1; i--) r.l[i] = (a.l[i] & FP_MASK) + c[i - 1];
  r.l[0] = a.l[0] & FP_MASK;
  r.l[FP_NL - 1] = top;
  FP_TRK(r.lb = lb; r.vb = vb; r.nn = false;)
}

// Value reduction: subtract the multiple of p nearest to the value (quotient estimated from the top limb, whose unit
// 2^364 is p / 106,514, so the lazy lower limbs cannot disturb it) in one exact carry pass.
// Result: limbs 0..12 in [0, 2^28), signed top limb, value in (-0.52 p, 0.52 p).  Any input with limbs below 2^31 - 2^8.
BLS_FN void fp_reduce_body(fp& r, const fp& a) {
  const int32_t k = (int32_t)rintf((float)a.l[FP_NL - 1] * FP_PTOP_INV);
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    const int64_t x = (int64_t)(a.l[i] + c) - (int64_t)k * (int32_t)FP_P[i];
    if (i < FP_NL - 1) {
      r.l[i] = (int32_t)((uint32_t)x & FP_MASK);
      c = (int32_t)(x >> FP_LB);
    } else {
      r.l[i] = (int32_t)x;
    }
  }
  FP_TRK(r.lb = FP_LB_N; r.vb = 0.52; r.nn = true;)
}
BLS_FN void fp_reduce(fp& r, const fp& a) {
  FP_TRK(if (a.lb >= 2147483392.0) fp_trk_fail("fp_reduce limb < 2^31 - 2^8", a.lb, 0); if (a.vb > 120.0) fp_trk_fail("fp_reduce |value| < 120 p", a.vb, 0);)
  fp_reduce_body(r, a);
}
// the same pass on the exact limbs of a non-negative integer below 2^392 (fp_from_raw): limbs in [0, 2^28), so the quotient (at most
// 2^392 / p = 2,522) times a limb of p stays far inside 64 bits and the carries inside 32
BLS_FN void fp_reduce_shifted_raw(fp& r, const fp& a) {
  FP_TRK(if (a.lb > FP_LB_N) fp_trk_fail("fp_reduce_shifted_raw exact limbs", a.lb, 0); if (a.vb > 2522.0) fp_trk_fail("fp_reduce_shifted_raw value < 2^392", a.vb, 0);)
  fp_reduce_body(r, a);
}

// The value reduction of ka a + kb b (small integer coefficients, e.g. 3 t - 2 z of a compressed squaring) in ONE pass: the
// combination is formed on 64 bits inside the carry chain, so the caller needs neither the limb-wise doublings and additions nor a
// carry pass to keep them within 32 bits.  Same result contract as fp_reduce.  Inputs: |ka| a.lb + |kb| b.lb below 2^35 (the carry
// into the top limb then moves the quotient estimate by less than 128 units of p / 106,514), |ka| a.vb + |kb| b.vb <= 120.
BLS_FN void fp_reduce_lin2(fp& r, const fp& a, int ka, const fp& b, int kb) {
  FP_TRK(const double tlb = (ka < 0 ? -ka : ka) * a.lb + (kb < 0 ? -kb : kb) * b.lb, tvb = (ka < 0 ? -ka : ka) * a.vb + (kb < 0 ? -kb : kb) * b.vb;
         if (tlb >= 34359738368.0) fp_trk_fail("fp_reduce_lin2 |limb| < 2^35", tlb, 0); if (tvb > 120.0) fp_trk_fail("fp_reduce_lin2 |value| < 120 p", tvb, 0);)
  const int32_t k = (int32_t)rintf((float)(ka * a.l[FP_NL - 1] + kb * b.l[FP_NL - 1]) * FP_PTOP_INV);
#if defined(__HIP_DEVICE_COMPILE__)
  // the coefficients as opaque scalars: three multiply-adds per limb; left to itself the compiler strength-reduces 3 a and 2 b into
  // 64-bit shifts, additions and borrow chains (seven instructions per limb instead of three)
  asm volatile("" : "+s"(ka));
  asm volatile("" : "+s"(kb));
#endif
  const int32_t nk = -k;
  int64_t c = 0;                                  // (the carry stays on 64 bits: it is the addend of the next limb's multiply-add chain)
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    const int64_t x = (int64_t)a.l[i] * ka + ((int64_t)b.l[i] * kb + ((int64_t)nk * (int32_t)FP_P[i] + c));
    if (i < FP_NL - 1) {
      r.l[i] = (int32_t)((uint32_t)x & FP_MASK);
      c = x >> FP_LB;
    } else {
      r.l[i] = (int32_t)x;
    }
  }
  FP_TRK(r.lb = FP_LB_N; r.vb = 0.52; r.nn = true;)
}

// The representative in [0, p) with exact limbs.
BLS_FN void fp_canon(fp& r, const fp& a) {
  fp t;
  fp_reduce(t, a);
  const int32_t m = t.l[FP_NL - 1] < 0 ? -1 : 0;   // negative -> add p
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    const int32_t x = t.l[i] + ((int32_t)FP_P[i] & m) + c;
    if (i < FP_NL - 1) {
      r.l[i] = x & FP_MASK;
      c = x >> FP_LB;
    } else {
      r.l[i] = x;
    }
  }
  FP_TRK(r.lb = FP_LB_N; r.vb = 1.0; r.nn = true;)
}

BLS_FN bool fp_is_zero(const fp& a) {
  fp t;
  fp_reduce(t, a);   // the only multiple of p in (-0.52 p, 0.52 p) is 0, and exact limbs represent it uniquely
  int32_t o = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) o |= t.l[i];
  return o == 0;
}

BLS_FN bool fp_eq(const fp& a, const fp& b) {
  fp d;
  fp_sub(d, a, b);
  return fp_is_zero(d);
}

// ---- Montgomery multiplication ---------------------------------------------------------------------
// Product scanning, R = 2^392:  r = REDC(a b)  or, with STREAMS == 2,  r = REDC(a b + c d)  in one pass (one reduction
// for two products: the Fp2 product of tower_split.cuh).  Every partial product is one 32 x 32 + 64 multiply-add into a
// signed 64-bit column accumulator; the column sum stays below 2^63 under the limb bounds stated at the top.
// STREAMS == 0: squaring, r = REDC(a a): the off-diagonal products are taken once against a doubled operand
// (105 multiply-adds instead of 196 for the product part; same column sums as the general form).
// Measured and rejected (round 3, tools/ubench/ubench3.hip form 4, profiles/r03_ubench3_leaf_forms.txt): every multiply-add as an
// opaque instruction in ONE accumulator chain.  The compiler, left to itself, splits the chain into several accumulators and
// joins them with 64-bit additions (28 v_lshl_add_u64 and 44 moves per fused pass, 113-130 registers); the single chain needs
// 60 registers and 8 % fewer cycles when two waves keep a SIMD busy all the time (6,069 -> 5,589 cycles per fused pass), but
// back-to-back dependent v_mad_i64_i32 need wait states (one s_nop per link), a wave that runs alone is 44 % slower (3,470 ->
// 5,002 cycles), and in the kernels -- where a wave's partner is parked in s_waitcnt 10-15 % of the time -- every kernel lost:
// k_prepare 1.58 -> 2.25 ms (one wave per SIMD), k_millerf2s 6.80 -> 7.39, k_finalexp2s 9.46 -> 10.09, k_lines2s 3.60 -> 3.85.
#define FP_MADI(acc, x, y) acc += (int64_t)(x) * (y)
#define FP_MADU(acc, x, y) acc += (int64_t)(x) * (int32_t)(y)
template <int STREAMS>
BLS_FN void fp_redc_products(fp& r, const fp& a, const fp& b, const fp& c, const fp& d) {
#if defined(BLS_TRACK_BOUNDS) && !defined(__HIPCC__)
  {
    const double prod = a.lb * b.lb + (STREAMS == 2 ? c.lb * d.lb : 0.0);
    if (14.0 * prod + 14.0 * 72057594037927936.0 + 68719476736.0 >= 9223372036854775808.0) fp_trk_fail("column sum < 2^63", a.lb * b.lb, STREAMS == 2 ? c.lb * d.lb : 0.0);
    const double vprod = a.vb * b.vb + (STREAMS == 2 ? c.vb * d.vb : 0.0);
    if (vprod > 256.0) fp_trk_fail("REDC input |a||b| + |c||d| <= 256 p^2", a.vb * b.vb, STREAMS == 2 ? c.vb * d.vb : 0.0);
  }
#endif
  int64_t acc = 0;
  int32_t m[FP_NL];
  int32_t t[FP_NL];
  int32_t a2[FP_NL];
  if (STREAMS == 0) {
#pragma unroll
    for (int i = 0; i < FP_NL; i++) a2[i] = a.l[i] + a.l[i];
  }
#pragma unroll
  for (int k = 0; k < 2 * FP_NL - 1; k++) {
    const int lo = k > FP_NL - 1 ? k - (FP_NL - 1) : 0, hi = k < FP_NL - 1 ? k : FP_NL - 1;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (STREAMS == 0) {
        if (2 * i < k) FP_MADI(acc, a.l[i], a2[k - i]);
        else if (2 * i == k) FP_MADI(acc, a.l[i], a.l[i]);
      } else {
        FP_MADI(acc, a.l[i], b.l[k - i]);
        if (STREAMS == 2) FP_MADI(acc, c.l[i], d.l[k - i]);
      }
    }
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (k < FP_NL && i == k) continue;  // m[k] is not known yet
      FP_MADU(acc, m[i], FP_P[k - i]);
    }
    if (k < FP_NL) {
      m[k] = (int32_t)(((uint32_t)acc * FP_N0INV) & FP_MASK);
      FP_MADU(acc, m[k], FP_P[0]);
      acc >>= FP_LB;
    } else {
      t[k - FP_NL] = (int32_t)((uint32_t)acc & FP_MASK);
      acc >>= FP_LB;
    }
  }
  t[FP_NL - 1] = (int32_t)acc;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = t[i];
  FP_TRK(r.lb = FP_LB_N; r.vb = 1.125; r.nn = true;)
}

// A WHOLE Fp2 product on one lane by Karatsuba (round 4, second session; tools/ubench/ubench4.hip form 3, measured there: 22.8 G Fp2
// products/s against 19.9 for the lane-split pair of fused passes):  (a0 + a1 u)(b0 + b1 u) = r0 + r1 u with
//     r0 = REDC(a0 b0 - a1 b1)            r1 = REDC((a0 + a1)(b0 + b1) - a0 b0 - a1 b1)
// Three product streams and two interleaved reduction chains: 3 x 196 + 2 x 196 = 980 multiply-adds (the lane-split pair: 2 x 588) and no
// partner exchange.  Column bound: the second accumulator holds 14 (|a0|+|a1|)(|b0|+|b1|) + 2 * 14 |a1||b1| + the reduction's 14 * 2^56 + a
// carry: below 2^63 for operands with limbs up to 2^28 + 2^8 (7 * 2^59.81 = 2^62.62), i.e. NORMALISED operands.
// fp_lin4: r = ka a + kb b + kc c + kd d limb-wise, small integer coefficients known at compile time (exact on the integers).  When every
// operand is a REDC / fp_reduce output (limbs 0..12 in [0, 2^28): the tracker's `nn`) the result's limbs lie between -N 2^28 and P 2^28
// for the sums N, P of the negative and positive coefficients -- the tracker then bounds them by the larger of the two instead of the
// sum of magnitudes (a0 - b0 + b1 stays below 2^29, not 3 * 2^28): what lets the Karatsuba pass below take sums of differences
// without a carry pass.
BLS_FN void fp_lin4(fp& r, const fp& a, int ka, const fp& b, int kb, const fp& c, int kc, const fp& d, int kd) {
#if defined(BLS_TRACK_BOUNDS) && !defined(__HIPCC__)
  const int ks[4] = {ka, kb, kc, kd};
  const fp* xs[4] = {&a, &b, &c, &d};
  double mag = 0, pos = 0, neg = 0, vb = 0;
  bool all_nn = true;
  for (int j = 0; j < 4; j++) {
    if (!ks[j]) continue;
    const double k = ks[j] < 0 ? -ks[j] : ks[j];
    mag += k * xs[j]->lb;
    vb += k * xs[j]->vb;
    if (ks[j] > 0) pos += k * xs[j]->lb; else neg += k * xs[j]->lb;
    all_nn = all_nn && xs[j]->nn;
  }
  const double lb = all_nn ? fmax(pos, neg) : mag;
  if (lb >= 2147483648.0) fp_trk_fail("fp_lin4 limb < 2^31", lb, mag);
#endif
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = ka * a.l[i] + kb * b.l[i] + kc * c.l[i] + kd * d.l[i];
  FP_TRK(r.lb = lb; r.vb = vb; r.nn = false;)
}
// the pass proper: operands prepared by the caller -- a0, na1 = -a1, b0, b1, s = a0 + a1, t = b0 + b1 (each with its own tracked bound).
// Returns r0 = REDC(a0 b0 - a1 b1) = c0 and rs = REDC((a0 + a1)(b0 + b1) - 2 a1 b1) = c1 + c0: with V' = sum (-a1) b1 in a fresh accumulator per
// column, a0 b0 chained onto r0's accumulator and (a0 + a1)(b0 + b1) onto rs's, the two chains take V' and 2 V' -- two 64-bit additions per
// column, where forming c1 = W - U - V itself takes U and V' apart and five; the caller subtracts r0 from rs on the reduced limbs (or
// folds that into the linear combination it forms anyway).
BLS_FN void fp2_kara_core(fp& r0, fp& rs, const fp& a0, const fp& na1, const fp& b0, const fp& b1, const fp& s_, const fp& t_) {
#if defined(BLS_TRACK_BOUNDS) && !defined(__HIPCC__)
  {
    const double uv = a0.lb * b0.lb + na1.lb * b1.lb, w = s_.lb * t_.lb + 2.0 * na1.lb * b1.lb;
    if (14.0 * fmax(uv, w) + 14.0 * 72057594037927936.0 + 68719476736.0 >= 9223372036854775808.0) fp_trk_fail("kara column sum < 2^63", uv, w);
    if (a0.vb * b0.vb + na1.vb * b1.vb > 256.0 || s_.vb * t_.vb + 2.0 * na1.vb * b1.vb > 256.0) fp_trk_fail("kara REDC input <= 256 p^2", a0.vb * b0.vb + na1.vb * b1.vb, s_.vb * t_.vb + 2.0 * na1.vb * b1.vb);
  }
#endif
  int32_t s[FP_NL], t[FP_NL], n1[FP_NL];
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    s[i] = s_.l[i];
    t[i] = t_.l[i];
    n1[i] = na1.l[i];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(s[i]));
    asm volatile("" : "+v"(t[i]));
    asm volatile("" : "+v"(n1[i]));
#endif
  }
  int64_t acc0 = 0, acc1 = 0;
  int32_t m0[FP_NL], m1[FP_NL], t0[FP_NL], t1[FP_NL];
#pragma unroll
  for (int k = 0; k < 2 * FP_NL - 1; k++) {
    const int lo = k > FP_NL - 1 ? k - (FP_NL - 1) : 0, hi = k < FP_NL - 1 ? k : FP_NL - 1;
    int64_t Vn = 0;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      FP_MADI(acc0, a0.l[i], b0.l[k - i]);
      FP_MADI(Vn, n1[i], b1.l[k - i]);
    }
#pragma unroll
    for (int i = lo; i <= hi; i++) FP_MADI(acc1, s[i], t[k - i]);
    acc0 += Vn;
    acc1 += 2 * Vn;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if (k < FP_NL && i == k) continue;
      FP_MADU(acc0, m0[i], FP_P[k - i]);
      FP_MADU(acc1, m1[i], FP_P[k - i]);
    }
    if (k < FP_NL) {
      m0[k] = (int32_t)(((uint32_t)acc0 * FP_N0INV) & FP_MASK);
      m1[k] = (int32_t)(((uint32_t)acc1 * FP_N0INV) & FP_MASK);
      FP_MADU(acc0, m0[k], FP_P[0]);
      FP_MADU(acc1, m1[k], FP_P[0]);
      acc0 >>= FP_LB;
      acc1 >>= FP_LB;
    } else {
      t0[k - FP_NL] = (int32_t)((uint32_t)acc0 & FP_MASK);
      t1[k - FP_NL] = (int32_t)((uint32_t)acc1 & FP_MASK);
      acc0 >>= FP_LB;
      acc1 >>= FP_LB;
    }
  }
  t0[FP_NL - 1] = (int32_t)acc0;
  t1[FP_NL - 1] = (int32_t)acc1;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    r0.l[i] = t0[i];
    rs.l[i] = t1[i];
  }
  FP_TRK(r0.lb = FP_LB_N; r0.vb = 1.125; r0.nn = true; rs.lb = FP_LB_N; rs.vb = 1.125; rs.nn = true;)
}
// (a0 + a1 u)(b0 + b1 u) for operands as they are: r0 = c0, rs = c1 + c0
BLS_FN void fp2_kara_products(fp& r0, fp& rs, const fp& a0, const fp& a1, const fp& b0, const fp& b1) {
  fp n1, s, t;
  fp_neg(n1, a1);
  fp_add(s, a0, a1);
  fp_add(t, b0, b1);
  fp2_kara_core(r0, rs, a0, n1, b0, b1, s, t);
}
// (a - b)(a - xi b) for REDUCED a, b (xi = 1 + u):  a - xi b = (a0 - b0 + b1) + (a1 - b0 - b1) u.  The operands' limbs lie in [0, 2^28),
// so every limb of these differences stays inside (-2^29, 2^29) -- a0 - b0 and a1 - b1 inside +-2^28 -- without a carry pass, and so do
// the two sums the Karatsuba pass takes, (a0 + a1) - (b0 + b1) and (a0 + a1) - 2 b0: 14 (2^58 + 2^58) + 14 * 2^56 + 2^36 = 126 * 2^56 + 2^36
// < 2^63 per column (fp_lin4's sign-aware bound is what proves it on the host; with the sum form (a + b)(a + xi b) the operands reach
// 3 * 2^28 and the pass needs four carry passes first).  Half of an Fp4 squaring: a^2 + xi b^2 = (a - b)(a - xi b) + (1 + xi) a b.
BLS_FN void fp2_kara_diffs(fp& r0, fp& rs, const fp& a0, const fp& a1, const fp& b0, const fp& b1) {      // r0 = c0, rs = c1 + c0
  fp x0, nx1, y0, y1, s, t;
  fp_lin4(x0, a0, 1, b0, -1, b0, 0, b0, 0);
  fp_lin4(nx1, b1, 1, a1, -1, a1, 0, a1, 0);        // -(a1 - b1)
  fp_lin4(y0, a0, 1, b1, 1, b0, -1, b0, 0);
  fp_lin4(y1, a1, 1, b0, -1, b1, -1, b1, 0);
  fp_lin4(s, a0, 1, a1, 1, b0, -1, b1, -1);
  fp_lin4(t, a0, 1, a1, 1, b0, -2, b0, 0);
  fp2_kara_core(r0, rs, x0, nx1, y0, y1, s, t);
}

#if defined(BLS_COUNT_FPMUL)
extern "C" { uint64_t g_fpmul_halves = 0; }   // tools/count_fpmul.py: host-side census, in half multiplications
#define FP_COUNT(n) g_fpmul_halves += (n)
#else
#define FP_COUNT(n)
#endif

#if defined(__HIPCC__)
// The multiplication bodies (about 520 / 770 instructions) are ONE non-inlined leaf each per translation unit whose 28
// operand limbs and 14 result limbs travel in VGPRs (ext_vector arguments: clang passes aggregates above 16 dwords
// through memory, which put every operand of every product into scratch, profiles/r01_pmc_before_leafcall.txt).
// Everything above is inlined so that values stay in registers between calls.
typedef int32_t i32x14 __attribute__((ext_vector_type(14)));
#define FP_OPAQUE(x) asm volatile("" : "+v"(x))   // keeps the sign extension next to the multiply: v_mad_i64_i32 selects
__device__ __noinline__ i32x14 fp_mul_leaf(i32x14 av, i32x14 bv) {
  fp a, b, r;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    a.l[i] = av[i];
    b.l[i] = bv[i];
    FP_OPAQUE(a.l[i]);
    FP_OPAQUE(b.l[i]);
  }
  fp_redc_products<1>(r, a, b, a, b);
  i32x14 o;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) o[i] = r.l[i];
  return o;
}
__device__ __noinline__ i32x14 fp_sqr_leaf(i32x14 av) {
  fp a, r;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    a.l[i] = av[i];
    FP_OPAQUE(a.l[i]);
  }
  fp_redc_products<0>(r, a, a, a, a);
  i32x14 o;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) o[i] = r.l[i];
  return o;
}
BLS_FN void fp_sqr(fp& r, const fp& a) {
  i32x14 x;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) x[i] = a.l[i];
  i32x14 o = fp_sqr_leaf(x);
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = o[i];
}
BLS_FN void fp_mul(fp& r, const fp& a, const fp& b) {
  i32x14 x, y;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    x[i] = a.l[i];
    y[i] = b.l[i];
  }
  i32x14 o = fp_mul_leaf(x, y);
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = o[i];
}
// ---- lane-split Fp2 product leaf (tower_split.cuh): each lane of an adjacent pair passes ITS component of a and b
// (28 dwords in VGPRs, nothing on the stack); the leaf fetches the partner's components by DPP quad_perm [1,0,3,2] and
// computes, in one fused two-product Montgomery pass:
//     even lane: REDC(a0 b0 + (-a1) b1) = c0          odd lane: REDC(a0 b1 + a1 b0) = c1
__device__ __forceinline__ int32_t dpp_swap(int32_t x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }
__device__ __noinline__ i32x14 fp2_mul_split_leaf(i32x14 av, i32x14 bv) {
  const bool hi = (threadIdx.x & 1u) != 0;
  fp b, pb, x0, x1, r;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    const int32_t pa = dpp_swap(av[i]);
    b.l[i] = bv[i];
    pb.l[i] = dpp_swap(bv[i]);
    x0.l[i] = hi ? pa : av[i];     // a0 on both lanes
    x1.l[i] = hi ? av[i] : -pa;    // even: -a1, odd: a1
    FP_OPAQUE(b.l[i]);
    FP_OPAQUE(pb.l[i]);
    FP_OPAQUE(x0.l[i]);
    FP_OPAQUE(x1.l[i]);
  }
  fp_redc_products<2>(r, x0, b, x1, pb);
  i32x14 o;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) o[i] = r.l[i];
  return o;
}
BLS_FN void fp2_mul_split(fp& r, const fp& a, const fp& b) {
  i32x14 x, y;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    x[i] = a.l[i];
    y[i] = b.l[i];
  }
  i32x14 o = fp2_mul_split_leaf(x, y);
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = o[i];
}
#else
BLS_FN void fp_mul(fp& r, const fp& a, const fp& b) {
  FP_COUNT(2);
  fp_redc_products<1>(r, a, b, a, b);
}
// host twin of the one-lane Karatsuba Fp2 product (tower_split.cuh's host emulation of the compressed squarings)
BLS_FN void fp2_mul_kara(fp& r0, fp& r1, const fp& a0, const fp& a1, const fp& b0, const fp& b1) {
  FP_COUNT(5);  // three product streams + two reductions = 2.5 multiplications
  fp rs;
  fp2_kara_products(r0, rs, a0, a1, b0, b1);
  fp_lin4(r1, rs, 1, r0, -1, r0, 0, r0, 0);
}
BLS_FN void fp2_mul_kara_diffs(fp& r0, fp& r1, const fp& a0, const fp& a1, const fp& b0, const fp& b1) {
  FP_COUNT(5);
  fp rs;
  fp2_kara_diffs(r0, rs, a0, a1, b0, b1);
  fp_lin4(r1, rs, 1, r0, -1, r0, 0, r0, 0);
}
BLS_FN void fp_sqr(fp& r, const fp& a) {
  FP_COUNT(2);
  fp_redc_products<0>(r, a, a, a, a);
}
// REDC(a b + c d): host twin of the lane-split product (tower_split.cuh's host emulation)
BLS_FN void fp_dotp2(fp& r, const fp& a, const fp& b, const fp& c, const fp& d) {
  FP_COUNT(3);  // two product streams + one reduction = 1.5 multiplications; two of these make one Fp2 product
  fp_redc_products<2>(r, a, b, c, d);
}
#endif

// ---- conversions ---------------------------------------------------------------------------------
// 12 little-endian 32-bit words of a plain integer < 2^384  ->  limbs of the same integer
BLS_FN void fp_set_words(fp& r, const uint32_t* w) {
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    const int bit = FP_LB * i, j = bit >> 5, s = bit & 31;
    uint64_t x = w[j];
    if (j + 1 < 12) x |= (uint64_t)w[j + 1] << 32;
    r.l[i] = (int32_t)((uint32_t)(x >> s) & FP_MASK);
  }
  FP_TRK(r.lb = FP_LB_N; r.vb = 9.85; r.nn = true;)   // 2^384 / p
}
// canonical limbs (fp_canon output) -> 12 little-endian 32-bit words
BLS_FN void fp_get_words(uint32_t* w, const fp& a) {
#pragma unroll
  for (int j = 0; j < 12; j++) {
    const int bit = 32 * j, i = bit / FP_LB, s = bit % FP_LB;
    uint64_t x = (uint64_t)(uint32_t)a.l[i] >> s;
    if (i + 1 < FP_NL) x |= (uint64_t)(uint32_t)a.l[i + 1] << (FP_LB - s);
    if (i + 2 < FP_NL) x |= (uint64_t)(uint32_t)a.l[i + 2] << (2 * FP_LB - s);
    w[j] = (uint32_t)x;
  }
}
// plain integer -> Montgomery, and back (canonical integer limbs)
BLS_FN void fp_to_mont(fp& r, const fp& a) {
  fp r2;
  fp_load(r2, FP_R2);
  fp_mul(r, a, r2);
}
BLS_FN void fp_from_mont(fp& r, const fp& a) {
  fp one, t;
  fp_load(one, FP_INT_ONE);
  fp_mul(t, a, one);
  fp_canon(r, t);
}
// caller format: blst's Montgomery words (R = 2^384).  The internal form is the same residue times 2^8 (R = 2^392): the limbs of the
// raw integer SHIFTED LEFT BY EIGHT BITS -- it stays below 2^392, fourteen limbs exactly -- and one value reduction.  No
// multiplication (round 4; rounds 1-3 multiplied by 2^400 mod p: ~530 instructions per element against ~75 -- 6 of the 49
// multiplication-equivalents per key of a 2^20-key sum).  The reduction's quotient estimate reads the top limb (below 2^28 here: the
// float conversion is off by at most 2^4 units of p / 106,514) and its carry pass is exact for any quotient below 2^31 / 2^28, so
// the 256 p (2,522 p for a caller's non-canonical words) of the shifted integer are as good as fp_reduce's usual few dozen.
BLS_FN void fp_from_raw(fp& r, const uint32_t* w) {
  fp t;
  t.l[0] = (int32_t)((w[0] << 8) & FP_MASK);
#pragma unroll
  for (int i = 1; i < FP_NL; i++) {
    const int bit = FP_LB * i - 8, j = bit >> 5, sh = bit & 31;
    uint64_t x = w[j];
    if (j + 1 < 12) x |= (uint64_t)w[j + 1] << 32;
    t.l[i] = (int32_t)((uint32_t)(x >> sh) & FP_MASK);
  }
  FP_TRK(t.lb = FP_LB_N; t.vb = 2522.0; t.nn = true;)   // 2^392 / p
  fp_reduce_shifted_raw(r, t);
}
// (the multiplication form, kept for the host-side cross-check of the shifted one: tests/hostsim hs_from_raw_forms_agree)
BLS_FN void fp_from_raw_mul(fp& r, const uint32_t* w) {
  fp t, k;
  fp_set_words(t, w);
  fp_load(k, FP_C400);
  fp_mul(r, t, k);
}
// ... and back: the caller's residue is the internal one divided by 2^8 modulo p.  Exactly, without a multiplication (round 4): T = the
// canonical integer in [0, p); the multiple m p with m = -T p^-1 mod 2^8 clears T's low eight bits (one digit of a Montgomery
// reduction in radix 2^8); (T + m p) / 2^8 is below p + p / 256, so one conditional subtraction of p gives the canonical words --
// the very words the multiplication by 2^384 mod p of rounds 1-3 produced (fp_to_raw_mul, cross-checked in tests/hostsim).
BLS_FN void fp_to_raw(uint32_t* w, const fp& a) {
  fp t, v, d;
  fp_canon(t, a);
  const int32_t m = (int32_t)(((uint32_t)t.l[0] * (FP_N0INV & 0xffu)) & 0xffu);
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {             // U = T + m p on exact limbs (the top limb takes what is left)
    const int64_t x = (int64_t)t.l[i] + (int64_t)m * (int32_t)FP_P[i] + c;
    if (i < FP_NL - 1) {
      t.l[i] = (int32_t)((uint32_t)x & FP_MASK);
      c = x >> FP_LB;
    } else {
      t.l[i] = (int32_t)x;
    }
  }
#pragma unroll
  for (int i = 0; i < FP_NL; i++)               // V = U / 2^8: limb i takes its own upper 20 bits and the low 8 of limb i + 1
    v.l[i] = (int32_t)(((uint32_t)t.l[i] >> 8) | (i + 1 < FP_NL ? ((uint32_t)t.l[i + 1] & 0xffu) << 20 : 0u));
  int32_t b = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {             // D = V - p with borrows
    const int32_t x = v.l[i] - (int32_t)FP_P[i] + b;
    if (i < FP_NL - 1) {
      d.l[i] = x & (int32_t)FP_MASK;
      b = x >> FP_LB;
    } else {
      d.l[i] = x;
    }
  }
  const bool ge = d.l[FP_NL - 1] >= 0;          // V >= p
#pragma unroll
  for (int i = 0; i < FP_NL; i++) v.l[i] = ge ? d.l[i] : v.l[i];
  FP_TRK(v.lb = FP_LB_N; v.vb = 1.0; v.nn = true;)
  fp_get_words(w, v);
}
BLS_FN void fp_to_raw_mul(uint32_t* w, const fp& a) {
  fp t, k;
  fp_load(k, FP_C384);
  fp_mul(t, a, k);
  fp_canon(t, t);
  fp_get_words(w, t);
}

// a^e for a public exponent given as little-endian 32-bit words (same for every lane: no divergence).
// Sliding windows of up to POW_WINDOW bits that end in a set bit, over the odd powers a, a^3 .. a^(2^W - 1): for the 379-bit
// (p-3)/4 that is 375 squarings + 78 multiplications + 8 for the table (fixed 4-bit digits: 376 + 92 + 14).
#define POW_WINDOW 4   // fp_pow's switch over the table assumes 2^(POW_WINDOW - 1) = 8 entries
// bits of a little-endian exponent through a 64-bit view of words cw and cw - 1: one load per word instead of one per bit (the
// scan is uniform scalar code, but a load per bit costs as much as the multiplications the windows save)
struct pow_bits {
  const uint32_t* e;
  int cw;
  uint64_t pair;
};
BLS_FN void pow_seek(pow_bits& x, int i) {          // before reading bits i, i - 1 .. i - 31
  if ((i >> 5) != x.cw) {
    x.cw = i >> 5;
    x.pair = ((uint64_t)x.e[x.cw] << 32) | (x.cw ? x.e[x.cw - 1] : 0u);
  }
}
BLS_FN uint32_t pow_bit(const pow_bits& x, int k) { return (uint32_t)(x.pair >> (k - 32 * x.cw + 32)) & 1u; }
// the window that starts at set bit i: its lowest bit j (also set) and its value
BLS_FN int pow_window(const pow_bits& x, int i, int width, uint32_t& value) {
  int j = i - width + 1 < 0 ? 0 : i - width + 1;
  while (!pow_bit(x, j)) j++;
  value = 0;
  for (int k = i; k >= j; k--) value = 2 * value + pow_bit(x, k);
  return j;
}
BLS_NOINLINE void fp_pow(fp& r, const fp& a, const uint32_t* e, int nbits) {
  fp tbl[1 << (POW_WINDOW - 1)], a2, acc;
  fp_norm(tbl[0], a);
  fp_sqr(a2, tbl[0]);
  for (int i = 1; i < (1 << (POW_WINDOW - 1)); i++) fp_mul(tbl[i], tbl[i - 1], a2);
  bool started = false;
  pow_bits x = {e, -1, 0};
  for (int i = nbits - 1; i >= 0;) {
    pow_seek(x, i);
    if (!pow_bit(x, i)) {
      if (started) fp_sqr(acc, acc);
      i--;
      continue;
    }
    uint32_t v;
    const int j = pow_window(x, i, POW_WINDOW, v);
    // the window value is the same on every lane (a public exponent): a uniform switch over STATIC table indices keeps the eight
    // odd powers in registers -- indexed with a run-time value the table lived in scratch, 14 loads per multiplication, and
    // k_prepare (two square-root chains per message) spent 30 % of its waves' time waiting for them
#define POW_PICK(op)                  \
  switch (v >> 1) {                   \
    case 0: op(tbl[0]); break;        \
    case 1: op(tbl[1]); break;        \
    case 2: op(tbl[2]); break;        \
    case 3: op(tbl[3]); break;        \
    case 4: op(tbl[4]); break;        \
    case 5: op(tbl[5]); break;        \
    case 6: op(tbl[6]); break;        \
    default: op(tbl[7]); break;       \
  }
#define POW_MUL(t) fp_mul(acc, acc, t)
#define POW_SET(t) acc = t
    if (started) {
      for (int k = i; k >= j; k--) fp_sqr(acc, acc);
      POW_PICK(POW_MUL)
    } else {
      POW_PICK(POW_SET)
      started = true;
    }
#undef POW_MUL
#undef POW_SET
#undef POW_PICK
    i = j - 1;
  }
  if (!started) fp_one(acc);
  r = acc;
}

// ---- inversion: constant-time "safegcd" (Bernstein-Yang divsteps) on the 28-bit limbs ---------------------------------
// a^(p-2) costs 380 squarings + ~100 multiplications (~210,000 instructions per lane); 40 batches of 28 divsteps with the
// 2x2 transition matrix of each batch applied to (f, g) and, modulo p, to (d, e) cost ~28,000, mostly carry-free 32-bit
// operations.  Data-independent control flow (every lane runs the same 1,120 divsteps: (49 * 381 + 57) / 17 = 1,101 is
// the published bound for 381-bit inputs), structure as in libsecp256k1's modinv32 with 28-bit limbs.
// Invariant: d x = f, e x = g (mod p); at the end g = 0, f = +-1 (f = p, d = 0 for x = 0), so x^-1 = +-d.
struct fp_divstep_mat {
  int32_t u, v, q, r;
};
#if defined(__HIPCC__)
#define FP_OPAQUE32(x) asm volatile("" : "+v"(x))   // keeps the operand a plain 32-bit value here: v_mad_i64_i32 selects
#else
#define FP_OPAQUE32(x)
#endif
BLS_FN int32_t fp_divsteps_28(int32_t eta, uint32_t f0, uint32_t g0, fp_divstep_mat& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 4
  for (int i = 0; i < FP_LB; i++) {
    uint32_t c1 = (uint32_t)(eta >> 31), c2 = 0u - (g & 1u);
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // negate f, u, v when eta < 0
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;                                                                // eta < 0 and g odd: swap roles
    eta = (int32_t)(((uint32_t)eta ^ c1) - (c1 + 1u));
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return eta;
}
// (f, g) <- t (f, g) / 2^28 (exact)
BLS_FN void fp_divstep_update_fg(int32_t* f, int32_t* g, const fp_divstep_mat& tt) {
  fp_divstep_mat t = tt;
  FP_OPAQUE32(t.u);
  FP_OPAQUE32(t.v);
  FP_OPAQUE32(t.q);
  FP_OPAQUE32(t.r);
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {   // masked limbs are known non-negative, which would turn the signed products into mixed ones
    FP_OPAQUE32(f[i]);
    FP_OPAQUE32(g[i]);
  }
  int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
  int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
  cf >>= FP_LB;
  cg >>= FP_LB;
#pragma unroll
  for (int i = 1; i < FP_NL; i++) {
    cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
    cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
    f[i - 1] = (int32_t)((uint32_t)cf & FP_MASK);
    g[i - 1] = (int32_t)((uint32_t)cg & FP_MASK);
    cf >>= FP_LB;
    cg >>= FP_LB;
  }
  f[FP_NL - 1] = (int32_t)cf;
  g[FP_NL - 1] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^28 mod p, keeping d, e in (-2p, p)
// (mask = 0 switches the multiples of p off: the same pass then IS the exact update of (f, g) -- tower_split.cuh fp_inv_pair runs it on a
// lane pair, (f, g) on one lane and (d, e) on the other)
BLS_FN void fp_divstep_update_de(int32_t* d, int32_t* e, const fp_divstep_mat& tt, int32_t mask = -1) {
  fp_divstep_mat t = tt;
  FP_OPAQUE32(t.u);
  FP_OPAQUE32(t.v);
  FP_OPAQUE32(t.q);
  FP_OPAQUE32(t.r);
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    FP_OPAQUE32(d[i]);
    FP_OPAQUE32(e[i]);
  }
  int32_t pl[FP_NL];
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    pl[i] = (int32_t)FP_P[i];
    FP_OPAQUE32(pl[i]);
  }
  const int32_t sd = d[FP_NL - 1] >> 31, se = e[FP_NL - 1] >> 31;
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
  int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
  md -= (int32_t)((FP_PINV28 * (uint32_t)cd + (uint32_t)md) & FP_MASK);
  me -= (int32_t)((FP_PINV28 * (uint32_t)ce + (uint32_t)me) & FP_MASK);
  md &= mask;
  me &= mask;
  FP_OPAQUE32(md);
  FP_OPAQUE32(me);
  cd += (int64_t)pl[0] * md;
  ce += (int64_t)pl[0] * me;
  cd >>= FP_LB;
  ce >>= FP_LB;
#pragma unroll
  for (int i = 1; i < FP_NL; i++) {
    cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)pl[i] * md;
    ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)pl[i] * me;
    d[i - 1] = (int32_t)((uint32_t)cd & FP_MASK);
    e[i - 1] = (int32_t)((uint32_t)ce & FP_MASK);
    cd >>= FP_LB;
    ce >>= FP_LB;
  }
  d[FP_NL - 1] = (int32_t)cd;
  e[FP_NL - 1] = (int32_t)ce;
}
// The same batches for PUBLIC operands (everything on a verify path is public) when a lone lane inverts -- the row-wide engine's
// Fp12 inversion, one item per workgroup: trailing zeros of g are divided out in one step, and up to eight low bits of g are
// cancelled at once by a multiple of f (table of -f^-1 mod 256), as in libsecp256k1's modinv32 "var" variant; ~7 loop rounds
// per batch instead of 28, the same transition matrix.  Data-dependent trip counts: not for kernels with one item per lane.
BLS_CONST uint8_t FP_NEGINV256[128] = {
    0xff, 0x55, 0x33, 0x49, 0xc7, 0x5d, 0x3b, 0x11, 0x0f, 0xe5, 0xc3, 0x59, 0xd7, 0xed, 0xcb, 0x21,
    0x1f, 0x75, 0x53, 0x69, 0xe7, 0x7d, 0x5b, 0x31, 0x2f, 0x05, 0xe3, 0x79, 0xf7, 0x0d, 0xeb, 0x41,
    0x3f, 0x95, 0x73, 0x89, 0x07, 0x9d, 0x7b, 0x51, 0x4f, 0x25, 0x03, 0x99, 0x17, 0x2d, 0x0b, 0x61,
    0x5f, 0xb5, 0x93, 0xa9, 0x27, 0xbd, 0x9b, 0x71, 0x6f, 0x45, 0x23, 0xb9, 0x37, 0x4d, 0x2b, 0x81,
    0x7f, 0xd5, 0xb3, 0xc9, 0x47, 0xdd, 0xbb, 0x91, 0x8f, 0x65, 0x43, 0xd9, 0x57, 0x6d, 0x4b, 0xa1,
    0x9f, 0xf5, 0xd3, 0xe9, 0x67, 0xfd, 0xdb, 0xb1, 0xaf, 0x85, 0x63, 0xf9, 0x77, 0x8d, 0x6b, 0xc1,
    0xbf, 0x15, 0xf3, 0x09, 0x87, 0x1d, 0xfb, 0xd1, 0xcf, 0xa5, 0x83, 0x19, 0x97, 0xad, 0x8b, 0xe1,
    0xdf, 0x35, 0x13, 0x29, 0xa7, 0x3d, 0x1b, 0xf1, 0xef, 0xc5, 0xa3, 0x39, 0xb7, 0xcd, 0xab, 0x01};
BLS_FN int32_t fp_divsteps_28_var(int32_t eta, uint32_t f0, uint32_t g0, fp_divstep_mat& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
  int i = FP_LB;
  for (;;) {
    const int zeros = __builtin_ctz(g | (0xffffffffu << i));      // the sentinel bit stops the count at i
    g >>= zeros;
    u <<= zeros;
    v <<= zeros;
    eta -= zeros;
    i -= zeros;
    if (i == 0) break;
    if (eta < 0) {
      uint32_t tmp;
      eta = -eta;
      tmp = f; f = g; g = 0u - tmp;
      tmp = u; u = q; q = 0u - tmp;
      tmp = v; v = r; r = 0u - tmp;
    }
    // eta >= 0: cancel the bottom bits of g -- at most i of them (the batch ends there), at most eta + 1 (the sign flips then)
    const int limit = (eta + 1) > i ? i : (eta + 1);
    const uint32_t m = (0xffffffffu >> (32 - limit)) & 255u;
    const uint32_t w = (g * (uint32_t)FP_NEGINV256[(f >> 1) & 127u]) & m;
    g += f * w;
    q += u * w;
    r += v * w;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return eta;
}
BLS_NOINLINE void fp_inv_var(fp& r, const fp& a) {   // 0 -> 0
  fp x;
  fp_canon(x, a);
  int32_t f[FP_NL], g[FP_NL], d[FP_NL], e[FP_NL];
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    f[i] = (int32_t)FP_P[i];
    g[i] = x.l[i];
    d[i] = 0;
    e[i] = 0;
  }
  e[0] = 1;
  int32_t eta = -1;
  for (int it = 0; it < 40; it++) {
    fp_divstep_mat t;
    const uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << FP_LB), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << FP_LB);
    eta = fp_divsteps_28_var(eta, f0, g0, t);
    fp_divstep_update_de(d, e, t);
    fp_divstep_update_fg(f, g, t);
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < FP_NL; i++) nz |= g[i];
    if (nz == 0) break;                             // g = 0: the remaining batches would change nothing
  }
  const int32_t neg = f[FP_NL - 1] >> 31;
  fp y, k;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) y.l[i] = (d[i] ^ neg) - neg;
  FP_TRK(y.lb = FP_LB_N; y.vb = 2.0; y.nn = false;)
  fp_load(k, FP_R3);
  fp_mul(r, y, k);
}
#if defined(BLS_FP_INV_VAR)
BLS_FN void fp_inv(fp& r, const fp& a) { fp_inv_var(r, a); }
#else
BLS_NOINLINE void fp_inv(fp& r, const fp& a) {   // 0 -> 0
  fp x;
  fp_canon(x, a);                               // the integer a R mod p in [0, p), exact limbs
  int32_t f[FP_NL], g[FP_NL], d[FP_NL], e[FP_NL];
#pragma unroll
  for (int i = 0; i < FP_NL; i++) {
    f[i] = (int32_t)FP_P[i];
    g[i] = x.l[i];
    d[i] = 0;
    e[i] = 0;
  }
  e[0] = 1;
  int32_t eta = -1;
  for (int it = 0; it < 40; it++) {
    fp_divstep_mat t;
    const uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << FP_LB), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << FP_LB);
    eta = fp_divsteps_28(eta, f0, g0, t);
    fp_divstep_update_de(d, e, t);
    fp_divstep_update_fg(f, g, t);
  }
  // f = +-1 (or p for x = 0): the inverse of a R as an integer is sign(f) d; times R^3 / R -> a^-1 R
  const int32_t neg = f[FP_NL - 1] >> 31;
  fp y, k;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) y.l[i] = (d[i] ^ neg) - neg;
  FP_TRK(y.lb = FP_LB_N; y.vb = 2.0; y.nn = false;)
  fp_load(k, FP_R3);
  fp_mul(r, y, k);
}

#endif

// Legendre symbol test: a is a square (0 counts as square)
BLS_FN bool fp_is_square(const fp& a) {
  fp t, one;
  fp_pow(t, a, EXP_PM1D2, EXP_PM1D2_BITS);
  fp_one(one);
  return fp_eq(t, one) || fp_is_zero(a);
}

// sqrt for p = 3 mod 4; returns false when a is not a square
BLS_FN bool fp_sqrt(fp& r, const fp& a) {
  fp t, s, c;
  fp_pow(t, a, EXP_PM3D4, EXP_PM3D4_BITS);  // a^((p-3)/4)
  fp_mul(s, t, a);                           // a^((p+1)/4)
  fp_sqr(c, s);
  bool ok = fp_eq(c, a);  // before writing r: r may alias a
  r = s;
  return ok;
}

// canonical-integer predicates (argument in Montgomery form)
BLS_FN uint32_t fp_parity(const fp& a) {
  fp t;
  fp_from_mont(t, a);
  return (uint32_t)t.l[0] & 1u;
}

// a (Montgomery) as integer > (p-1)/2 ?
BLS_FN bool fp_lex_largest(const fp& a) {
  fp t;
  fp_from_mont(t, a);
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) c = ((int32_t)FP_PM1D2[i] - t.l[i] + c) >> FP_LB;
  return c < 0;  // (p-1)/2 - a < 0
}
