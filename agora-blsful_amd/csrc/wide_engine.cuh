// The row-wide engine: Fp12 arithmetic of ONE item on a 256-thread workgroup (four waves, one per SIMD of a CU, sixteen
// DPP rows), every Fp value living in LDS as sixteen words (limb j at word j, csrc/wide.cuh).  An operation is a table
// (csrc/wide_tables.cuh, generated and oracle-checked by tools/gen_wide_tables.py): product sub-rounds in which every row
// multiplies two short sums of values, then one linear phase in which up to sixteen rows combine products (and old values)
// into the new coefficients and reduce them.  A program -- e.g. the hard part of the final exponentiation, 363 steps -- is
// a table of (operation, destination array, operand arrays) words that one inlined interpreter loop walks, so the
// multiplier's per-lane constants stay in registers and every access is a plain LDS access.  Additions cost one
// instruction here and a product ~240, against ~500 plus a few hundred instructions of carries and selections per product
// round of the lane-pair form (coop.cuh) -- and on a lone wave time is instruction count.
#pragma once
#include "wide.cuh"
#include "wide_tables.cuh"

// value store (indices of 16-word values): the named arrays WV_F.. of wide_tables.cuh, then product scratch, then constants
#define WV_TMP 60
#define WV_CONST (WV_TMP + WIDE_MAX_TMP)
#define WV_COUNT (WV_CONST + 24)
#define WIDE_NPROD (sizeof(WIDE_PROD) / sizeof(wide_prod))
#define WIDE_NLIN (sizeof(WIDE_LIN) / sizeof(wide_lin))
#define WIDE_NOPS (sizeof(WIDE_OPS) / sizeof(wide_op))
#define WIDE_PROG_MAX 512

struct wide_lds {
  uint32_t V[WV_COUNT][16];
  wide_prod prod[WIDE_NPROD];     // the tables and the program are staged in LDS: an interpreter on a lone wave cannot
  wide_lin lin[WIDE_NLIN];        // afford a global-memory round trip per step
  wide_op ops[WIDE_NOPS];
  uint32_t prog[WIDE_PROG_MAX];
  int flag;
};

__device__ __forceinline__ void wide_stage(wide_lds& S, const uint32_t* prog, int prog_len) {
  const uint32_t *sp = (const uint32_t*)WIDE_PROD, *sl = (const uint32_t*)WIDE_LIN, *so = (const uint32_t*)WIDE_OPS;
  uint32_t *dp = (uint32_t*)S.prod, *dl = (uint32_t*)S.lin, *dq = (uint32_t*)S.ops;
  for (unsigned t = threadIdx.x; t < sizeof(WIDE_PROD) / 4; t += blockDim.x) dp[t] = sp[t];
  for (unsigned t = threadIdx.x; t < sizeof(WIDE_LIN) / 4; t += blockDim.x) dl[t] = sl[t];
  for (unsigned t = threadIdx.x; t < sizeof(WIDE_OPS) / 4; t += blockDim.x) dq[t] = so[t];
  for (int t = threadIdx.x; t < prog_len; t += blockDim.x) S.prog[t] = prog[t];
  for (int t = threadIdx.x; t < 24 * 16; t += blockDim.x) {    // Frobenius constants xi^(k (p^j - 1) / 6): value 12 (j - 1) + 2 k + component
    const int v = t >> 4, l = t & 15, j = v / 12, k = (v % 12) >> 1, comp = v & 1;
    S.V[WV_CONST + v][l] = l < FP_NL ? (j == 0 ? FROB1[k][comp * FP_NL + l] : FROB2[k][comp * FP_NL + l]) : 0u;
  }
}

// 64-bit column sums of a linear phase (magnitude below 2^40) -> normalised limbs of a value reduced to (-0.6 p, 0.6 p)
__device__ __forceinline__ wfp w_finish(int64_t acc, const wide_consts& K) {
  const int64_t c1 = acc >> FP_LB;
  const int32_t c1m = (int32_t)c1 & K.pass;
  const int64_t keep64 = K.lane < FP_NL - 1 ? (int64_t)FP_MASK : (K.lane == FP_NL - 1 ? (int64_t)-1 : (int64_t)0);
  const int32_t v1 = (int32_t)(acc & keep64) + w_shr<1>(c1m);
  // nearest multiple of p from the top limb (its unit is p / 106,513.6), subtracted on 64 bits
  const int32_t top = w_bcast<13>(v1);
  const int32_t k = __float2int_rn((float)top * (1.0f / 106513.57f));
  const int32_t pl = K.plo[0];                               // this lane's limb of p
  const int64_t t = (int64_t)v1 - (int64_t)k * pl;
  const int64_t c2 = t >> FP_LB;
  const int32_t c2m = (int32_t)c2 & K.pass;
  const int32_t v2 = (int32_t)(t & keep64) + w_shr<1>(c2m);
  return w_norm(v2, K);
}

// runs S.prog[0 .. prog_len): every step is dst = op(a, b) on value arrays (dst may alias an operand: products are staged in
// the scratch values and a linear row reads nothing that another row writes).  Call from all 256 threads.
__device__ __forceinline__ void wide_exec(wide_lds& S, int prog_len, const wide_consts& K) {
  const int row = (int)(threadIdx.x >> 4), lane = (int)(threadIdx.x & 15u);
  for (int pc = 0; pc < prog_len; pc++) {
    const uint32_t w = S.prog[pc];
    const wide_op op = S.ops[w & 0xffu];
    const uint32_t bd = (w >> 8) & 0xffu, ba = (w >> 16) & 0xffu, bb = op.b_is_const ? (uint32_t)WV_CONST : (op.b_is_a ? ba : (w >> 24));
    if (op.nsub) {
      // software pipeline: the operands of sub-round t + 1 (and the table entry of t + 2) are fetched before the product of
      // sub-round t runs -- products read only operand arrays and constants, never the scratch values they write
      wide_prod e = S.prod[op.prod_off + row];
      wfp a0 = (wfp)S.V[ba + e.a[0]][lane], a1 = (wfp)S.V[ba + e.a[1]][lane];
      wfp b0 = (wfp)S.V[bb + e.b[0]][lane], b1 = (wfp)S.V[bb + e.b[1]][lane];
      for (int t = 0; t < op.nsub; t++) {
        const wfp A = a0 * (int32_t)e.ca[0] + a1 * (int32_t)e.ca[1];
        const wfp B = b0 * (int32_t)e.cb[0] + b1 * (int32_t)e.cb[1];
        const uint32_t out = WV_TMP + e.out;
        if (t + 1 < op.nsub) {
          e = S.prod[op.prod_off + 16 * (t + 1) + row];
          a0 = (wfp)S.V[ba + e.a[0]][lane];
          a1 = (wfp)S.V[ba + e.a[1]][lane];
          b0 = (wfp)S.V[bb + e.b[0]][lane];
          b1 = (wfp)S.V[bb + e.b[1]][lane];
        }
        const wfp p = w_mul(A, B, K);
        S.V[out][lane] = (uint32_t)p;
      }
      __syncthreads();
    }
    if (row < op.nlin) {
      const wide_lin* L = &S.lin[op.lin_off + row];
      const int n = L->n;
      int64_t acc = 0;
      // six terms per step: their value loads are issued together (the unused ones carry coefficient zero)
      for (int i0 = 0; i0 < n; i0 += 6) {
        wfp v[6];
        int32_t k[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
          const uint32_t ix = L->idx[i0 + j];
          v[j] = (wfp)S.V[((ix & 0x80u) ? ba : (uint32_t)WV_TMP) + (ix & 0x7fu)][lane];
          k[j] = (int32_t)L->c[i0 + j];
        }
#pragma unroll
        for (int j = 0; j < 6; j++) acc += (int64_t)k[j] * v[j];
      }
      S.V[bd + L->out][lane] = (uint32_t)w_finish(acc, K);
    }
    __syncthreads();
  }
}
