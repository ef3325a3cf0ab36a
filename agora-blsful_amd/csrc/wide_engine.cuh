// The row-wide engine: Fp12 / twist-point arithmetic of ONE item on a 256-thread workgroup (four waves, one per SIMD of a
// CU, sixteen DPP rows), every Fp value living in LDS as sixteen words (limb j at word j, csrc/wide.cuh).  An operation is
// a table (csrc/wide_tables.cuh, generated and oracle-checked by tools/gen_wide_tables.py): product sub-rounds in which
// every row multiplies two short sums of values, then one linear phase in which rows combine products (and old values) into
// new values and reduce them.  A program -- the key's line coefficients, the Miller loop, the final exponentiation: ~850
// steps for one core_verify -- is a table of (operation, destination array, operand arrays) words that one inlined
// interpreter loop walks, so the multiplier's per-lane constants stay in registers and every access is a plain LDS access.
// Additions cost one instruction here and a product ~240, against ~500 plus a few hundred instructions of carries and
// selections per product round of the lane-pair form (coop.cuh) -- and on a lone wave time is instruction count.
#pragma once
#include "tower_split.cuh"
#include "wide.cuh"
#include "wide_tables.cuh"

#define WIDE_NPROD (sizeof(WIDE_PROD) / sizeof(wide_prod))
#define WIDE_NLIN (sizeof(WIDE_LIN) / sizeof(wide_lin))
#define WIDE_NOPS (sizeof(WIDE_OPS) / sizeof(wide_op))

struct wide_lds {
  uint32_t V[WV_COUNT][16];
  wide_prod prod[WIDE_NPROD];     // the tables and the program are staged in LDS: an interpreter on a lone wave cannot
  wide_lin lin[WIDE_NLIN];        // afford a global-memory round trip per step
  wide_op ops[WIDE_NOPS];
  uint32_t prog[2 * WIDE_PROG_MAX];
  int flag;
};

__device__ __forceinline__ void wide_stage(wide_lds& S, const uint32_t* prog, int prog_len) {
  const uint32_t *sp = (const uint32_t*)WIDE_PROD, *sl = (const uint32_t*)WIDE_LIN, *so = (const uint32_t*)WIDE_OPS;
  uint32_t *dp = (uint32_t*)S.prod, *dl = (uint32_t*)S.lin, *dq = (uint32_t*)S.ops;
  for (unsigned t = threadIdx.x; t < sizeof(WIDE_PROD) / 4; t += blockDim.x) dp[t] = sp[t];
  for (unsigned t = threadIdx.x; t < sizeof(WIDE_LIN) / 4; t += blockDim.x) dl[t] = sl[t];
  for (unsigned t = threadIdx.x; t < sizeof(WIDE_OPS) / 4; t += blockDim.x) dq[t] = so[t];
  for (int t = threadIdx.x; t < 2 * prog_len; t += blockDim.x) S.prog[t] = prog[t];
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

// T <- F^-1 on lane pair 0 of the workgroup with the lane-split tower code (tower_split.cuh): the one operation of a
// pairing that is neither a product table nor linear (one safegcd inversion inside)
__device__ __noinline__ void wide_inv_f(wide_lds& S) {
  if (threadIdx.x < 2) {
    const int hi = (int)threadIdx.x;
    fp12_t<hfp2> a, b;
    hfp2* ca[6] = {&a.c0.a0, &a.c1.a0, &a.c0.a1, &a.c1.a1, &a.c0.a2, &a.c1.a2};   // coefficient k of w^k in tower order
    hfp2* cb[6] = {&b.c0.a0, &b.c1.a0, &b.c0.a1, &b.c1.a1, &b.c0.a2, &b.c1.a2};
#pragma unroll
    for (int k = 0; k < 6; k++) w_load_local(ca[k]->v, S.V[WV_F + 2 * k + hi]);
    fp12_inv(b, a);
#pragma unroll
    for (int k = 0; k < 6; k++) w_store_local(S.V[WV_T + 2 * k + hi], cb[k]->v);
  }
}

// runs S.prog[0 .. prog_len): every step is dst = op(a, b) on value arrays (dst may alias an operand: products are staged in
// the scratch values and a linear row reads nothing that another row writes).  Call from all 256 threads.
__device__ __forceinline__ void wide_exec(wide_lds& S, int prog_len, const wide_consts& K) {
  const int row = (int)(threadIdx.x >> 4), lane = (int)(threadIdx.x & 15u);
  for (int pc = 0; pc < prog_len; pc++) {
    const uint32_t w0 = S.prog[2 * pc], w1 = S.prog[2 * pc + 1];
    const uint32_t opid = w0 & 0xffu;
    if (opid == WOP_INV) {
      wide_inv_f(S);
      __syncthreads();
      continue;
    }
    const wide_op op = S.ops[opid];
    const uint32_t bd = w0 >> 16, ba = w1 & 0xffffu, bb = op.b_is_const ? (uint32_t)WV_CONST : (op.b_is_a ? ba : (w1 >> 16));
    if (op.nsub) {
      wide_prod e = S.prod[op.prod_off + row];
      for (int t = 0; t < op.nsub; t++) {
        const wfp a0 = (wfp)S.V[ba + e.a[0]][lane], a1 = (wfp)S.V[ba + e.a[1]][lane];
        const wfp b0 = (wfp)S.V[bb + e.b[0]][lane], b1 = (wfp)S.V[bb + e.b[1]][lane];
        const wfp A = a0 * (int32_t)e.ca[0] + a1 * (int32_t)e.ca[1];
        const wfp B = b0 * (int32_t)e.cb[0] + b1 * (int32_t)e.cb[1];
        const uint32_t out = WV_TMP + e.out;
        if (t + 1 < op.nsub) e = S.prod[op.prod_off + 16 * (t + 1) + row];     // the next entry travels while this product runs
        S.V[out][lane] = (uint32_t)w_mul(A, B, K);
      }
      __syncthreads();
    }
    for (int r = row; r < op.nlin; r += 16) {
      const wide_lin* L = &S.lin[op.lin_off + r];
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
