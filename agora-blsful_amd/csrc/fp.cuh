// Fp arithmetic for BLS12-381 on gfx950: 12 x 32-bit limbs, Montgomery form, R = 2^384.
//
// The in-memory form is byte-identical to blst's 6 x u64 little-endian Montgomery limbs, which is what
// the reference's G1Projective/G2Projective hold, so RAW_PROJ buffers need no conversion.
// 32-bit limbs because the CDNA4 integer multiplier is 32 x 32 (+64) -> 64 (v_mad_u64_u32).
//
// This header also compiles as plain C++ (no HIP) for the host-side unit tests in tests/hostsim.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BLS_FN __device__ __forceinline__
#define BLS_NOINLINE __device__ __noinline__
#define BLS_CONST __device__ __constant__ const
#else
#define BLS_FN static inline
#define BLS_NOINLINE static __attribute__((noinline))
#define BLS_CONST static const
#endif

#include "consts.cuh"

struct fp {
  uint32_t l[12];
};

// ---- carry helpers -------------------------------------------------------------------------------
BLS_FN uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) {
#if defined(__clang__)
  unsigned co;
  uint32_t r = __builtin_addc(a, b, carry, &co);
  carry = co;
  return r;
#else
  uint64_t s = (uint64_t)a + b + carry;
  carry = (uint32_t)(s >> 32);
  return (uint32_t)s;
#endif
}

BLS_FN uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) {
#if defined(__clang__)
  unsigned bo;
  uint32_t r = __builtin_subc(a, b, borrow, &bo);
  borrow = bo;
  return r;
#else
  uint64_t s = (uint64_t)a - b - borrow;
  borrow = (uint32_t)(s >> 63);
  return (uint32_t)s;
#endif
}

// ---- basic ops -----------------------------------------------------------------------------------
BLS_FN void fp_load(fp& r, const uint32_t* c) {
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = c[i];
}

BLS_FN void fp_store(uint32_t* c, const fp& a) {
#pragma unroll
  for (int i = 0; i < 12; i++) c[i] = a.l[i];
}

BLS_FN void fp_zero(fp& r) {
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = 0;
}

BLS_FN void fp_one(fp& r) { fp_load(r, FP_ONE); }

BLS_FN bool fp_is_zero(const fp& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) o |= a.l[i];
  return o == 0;
}

BLS_FN bool fp_eq(const fp& a, const fp& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}

BLS_FN void fp_cmov(fp& r, const fp& a, bool c) {  // r = c ? a : r
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = c ? a.l[i] : r.l[i];
}

// r = a - p if a >= p else a   (a < 2p)
BLS_FN void fp_reduce_once(fp& r, const fp& a) {
  uint32_t d[12], bw = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) d[i] = subb32(a.l[i], FP_P[i], bw);
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = bw ? a.l[i] : d[i];
}

BLS_FN void fp_add(fp& r, const fp& a, const fp& b) {
  fp t;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) t.l[i] = addc32(a.l[i], b.l[i], c);
  fp_reduce_once(r, t);  // a + b < 2p < 2^384: no carry out
}

BLS_FN void fp_sub(fp& r, const fp& a, const fp& b) {
  uint32_t d[12], bw = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) d[i] = subb32(a.l[i], b.l[i], bw);
  uint32_t mask = 0u - bw, c = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = addc32(d[i], FP_P[i] & mask, c);
}

BLS_FN void fp_neg(fp& r, const fp& a) {
  uint32_t bw = 0, nz = 0;
  uint32_t d[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    nz |= a.l[i];
    d[i] = subb32(FP_P[i], a.l[i], bw);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = nz ? d[i] : 0u;
}

BLS_FN void fp_dbl(fp& r, const fp& a) { fp_add(r, a, a); }

// ---- Montgomery multiplication ---------------------------------------------------------------------
// Portable CIOS (interleaved; top limb of p < 2^31 so no extra carry word).  Used by the host-side unit
// tests and as the reference the asm form is checked against (tools/ubench).
BLS_FN void fp_mul_c(fp& r, const fp& a, const fp& b) {
  uint32_t t[12];
#pragma unroll
  for (int i = 0; i < 12; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const uint32_t bi = b.l[i];
    uint64_t A = (uint64_t)a.l[0] * bi + t[0];
    const uint32_t m = (uint32_t)A * FP_N0INV;
    uint64_t C = (uint64_t)m * FP_P[0] + (uint32_t)A;
    A >>= 32;
    C >>= 32;
#pragma unroll
    for (int j = 1; j < 12; j++) {
      A += (uint64_t)a.l[j] * bi + t[j];
      C += (uint64_t)m * FP_P[j] + (uint32_t)A;
      t[j - 1] = (uint32_t)C;
      A >>= 32;
      C >>= 32;
    }
    t[11] = (uint32_t)(A + C);
  }
  fp tt;
#pragma unroll
  for (int i = 0; i < 12; i++) tt.l[i] = t[i];
  fp_reduce_once(r, tt);
}

#if defined(__HIPCC__)
// gfx950: product-scanning form, one v_mad_u64_u32 + one v_addc_co_u32 per partial product.  Measured on
// MI355X (profiles/ubench_r01.txt): 58 G fp_mul/s chip-wide at 4 waves/SIMD vs 40 G for the C form.
#include "fp_mul_gfx950.inc"
// The multiplication body (about 720 instructions) is ONE non-inlined leaf per translation unit whose 24 operand
// limbs and 12 result limbs travel in VGPRs (scalar arguments: clang passes aggregates above 16 dwords through
// memory, which put every operand of every product into scratch: 80 GB of fabric traffic per 65,536-item Miller
// launch and waves parked in s_waitcnt 47 % of the time, profiles/r01_pmc_before_leafcall.txt).  Everything above
// fp_mul is inlined so that values stay in registers between calls.
struct fp_ret {
  uint32_t l[12];
};
__device__ __noinline__ fp_ret fp_mul_leaf(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6,
                                           uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t b0, uint32_t b1,
                                           uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8,
                                           uint32_t b9, uint32_t b10, uint32_t b11) {
  fp a = {{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11}}, b = {{b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11}}, r;
  fp_mul_asm(r, a, b);
  fp_ret o;
#pragma unroll
  for (int i = 0; i < 12; i++) o.l[i] = r.l[i];
  return o;
}
BLS_FN void fp_mul(fp& r, const fp& a, const fp& b) {
  fp_ret t = fp_mul_leaf(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], b.l[0], b.l[1],
                         b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11]);
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = t.l[i];
}
// ---- lane-split Fp2 product leaf (tower_split.cuh): each lane of an adjacent pair passes ITS component of a and b
// (24 dwords in VGPRs, nothing on the stack); the leaf fetches the partner's components by DPP quad_perm [1,0,3,2] and
// computes, in one fused two-product Montgomery pass (lazy reduction):
//     even lane: REDC(a0 b0 + (p - a1) b1) = c0          odd lane: REDC(a0 b1 + a1 b0) = c1
typedef uint32_t u32x12 __attribute__((ext_vector_type(12)));
__device__ __forceinline__ uint32_t dpp_swap(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); }
__device__ __noinline__ u32x12 fp2_mul_split_leaf(u32x12 av, u32x12 bv) {
  const bool hi = (threadIdx.x & 1u) != 0;
  fp a, b, pa, pb, npa, x0, x1, r;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    a.l[i] = av[i];
    b.l[i] = bv[i];
    pa.l[i] = dpp_swap(av[i]);
    pb.l[i] = dpp_swap(bv[i]);
  }
  fp_neg(npa, pa);
#pragma unroll
  for (int i = 0; i < 12; i++) {
    x0.l[i] = hi ? pa.l[i] : a.l[i];    // a0 on both lanes
    x1.l[i] = hi ? a.l[i] : npa.l[i];   // even: -a1, odd: a1
  }
  fp_dotp2_asm(r, x0, b, x1, pb);
  u32x12 o;
#pragma unroll
  for (int i = 0; i < 12; i++) o[i] = r.l[i];
  return o;
}
BLS_FN void fp2_mul_split(fp& r, const fp& a, const fp& b) {
  u32x12 x, y;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    x[i] = a.l[i];
    y[i] = b.l[i];
  }
  u32x12 o = fp2_mul_split_leaf(x, y);
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = o[i];
}
#else
#if defined(BLS_COUNT_FPMUL)
extern "C" { uint64_t g_fpmul_count = 0; }   // tools/count_fpmul.py: host-side instruction-mix census
BLS_FN void fp_mul(fp& r, const fp& a, const fp& b) { g_fpmul_count++; fp_mul_c(r, a, b); }
#else
BLS_FN void fp_mul(fp& r, const fp& a, const fp& b) { fp_mul_c(r, a, b); }
#endif
#endif

BLS_FN void fp_sqr(fp& r, const fp& a) { fp_mul(r, a, a); }

// plain integer -> Montgomery, and back
BLS_FN void fp_to_mont(fp& r, const fp& a) {
  fp r2;
  fp_load(r2, FP_R2);
  fp_mul(r, a, r2);
}

BLS_FN void fp_from_mont(fp& r, const fp& a) {
  fp one;
  fp_zero(one);
  one.l[0] = 1;
  fp_mul(r, a, one);
}

// a^e for a public exponent given as little-endian 32-bit words (same for every lane: no divergence).
// Fixed 4-bit windows: bits/4 * 4 squarings + one multiplication per non-zero digit + 14 for the table.
BLS_NOINLINE void fp_pow(fp& r, const fp& a, const uint32_t* e, int nbits) {
  fp tbl[16];
  fp_one(tbl[0]);
  tbl[1] = a;
  for (int i = 2; i < 16; i++) fp_mul(tbl[i], tbl[i - 1], a);
  fp acc;
  fp_one(acc);
  const int ndig = (nbits + 3) / 4;
  for (int d = ndig - 1; d >= 0; d--) {
    if (d != ndig - 1) {
      fp_sqr(acc, acc);
      fp_sqr(acc, acc);
      fp_sqr(acc, acc);
      fp_sqr(acc, acc);
    }
    const uint32_t dig = (e[d >> 3] >> ((d & 7) * 4)) & 15u;
    if (dig) fp_mul(acc, acc, tbl[dig]);
  }
  r = acc;
}

BLS_FN void fp_inv(fp& r, const fp& a) { fp_pow(r, a, EXP_PM2, EXP_PM2_BITS); }  // 0 -> 0

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
  return t.l[0] & 1;
}

// a (Montgomery) as integer > (p-1)/2 ?
BLS_FN bool fp_lex_largest(const fp& a) {
  fp t;
  fp_from_mont(t, a);
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) (void)subb32(FP_PM1D2[i], t.l[i], bw);
  return bw != 0;  // (p-1)/2 - a < 0
}
