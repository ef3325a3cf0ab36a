// Wave-cooperative pairing: ONE 64-lane wave per item, for the single-item tails of MultiSignature::verify and
// verify_secure (one core_verify after the big parallel part) and for small batches, where the per-item kernels leave
// the chip idle and one lane pair needs ~25 ms.
//
// An Fp12 value is held in LDS as its six Fp2 coefficients over the basis w^0..w^5 (w^6 = xi) -- the 2-3-2 tower's
// coefficients in the order c0.a0, c1.a0, c0.a1, c1.a1, c0.a2, c1.a2.  The 32 lane pairs of the wave each own one
// (or two) of the 36 coefficient products a_i b_j; a product is the fused two-product Montgomery pass of
// tower_split.cuh (even lane: real part, odd lane: imaginary part).  Six lane pairs then reduce the products along the
// anti-diagonals i + j = k (mod 6), multiplying the wrapped half by xi.  Squaring uses the 21 products i <= j, the
// sparse line multiplication 18.  Point arithmetic of the Miller loop (9 dependent Fp2 products per doubling) stays on
// lane pair 0, which runs the ordinary lane-split template code.
// Device only.
#pragma once
#include "tower_split.cuh"
#include "verify.cuh"

#define COOP_FP2_WORDS (2 * FP_NL)
struct coop_f12 {
  uint32_t c[6][COOP_FP2_WORDS];  // coefficient k: FP_NL words real part, then FP_NL words imaginary part
};
struct coop_shared {
  coop_f12 f, t, u, v, acc;
  uint32_t prod[36][COOP_FP2_WORDS];
  uint32_t line[2][3][COOP_FP2_WORDS];   // the line values of the two pairs of one Miller step
  uint32_t job[2][6][2][COOP_FP2_WORDS];  // operand pairs of the point-step products of the two pairs (coop_jobs)
  uint32_t res[2][6][COOP_FP2_WORDS];     // ... and their results
  int flag;
};

__device__ __forceinline__ int coop_pair() { return (int)(threadIdx.x >> 1); }
__device__ __forceinline__ void coop_ld(hfp2& r, const uint32_t* slot) {
  const uint32_t* p = slot + (lane_hi() ? FP_NL : 0);
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.v.l[i] = (int32_t)p[i];
}
__device__ __forceinline__ void coop_st(uint32_t* slot, const hfp2& a) {
  uint32_t* p = slot + (lane_hi() ? FP_NL : 0);
#pragma unroll
  for (int i = 0; i < FP_NL; i++) p[i] = (uint32_t)a.v.l[i];
}
// Anti-diagonal reduction: lane pair k < 6 builds  sum_{i+j = k} w P_ij  +  xi * sum_{i+j = k+6} w P_ij  from the staged
// products.  MODE 0: full product (P_ij at prod[6 i + j]);  1: squaring (prod index of i <= j, off-diagonal weight 2);
// 2: sparse line (prod[3 i + c], c = 0, 1, 2 for the line coefficients at w^0, w^2, w^3).
__device__ __forceinline__ int coop_sqr_slot(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }   // i <= j
template <int MODE>
__device__ __forceinline__ void coop_reduce(coop_shared& S, coop_f12& dst) {
  const int k = coop_pair();
  if (k < 6) {
    hfp2 lo, hiacc, p;
    fp2_zero(lo);
    fp2_zero(hiacc);
    if (MODE == 2) {
      const int jw[3] = {0, 2, 3};
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int i = (k - jw[c] + 6) % 6;             // i + jw[c] = k or k + 6
        coop_ld(p, S.prod[3 * i + c]);
        if (i + jw[c] == k) fp2_add(lo, lo, p);
        else fp2_add(hiacc, hiacc, p);
      }
    } else {
      for (int i = 0; i < 6; i++) {
        const int j = (k - i + 6) % 6;
        if (MODE == 1) {
          if (i > j) continue;                          // the pair (j, i) is visited as (i', j') with i' < j'
          coop_ld(p, S.prod[coop_sqr_slot(i, j)]);
          if (i != j) fp2_dbl(p, p);
        } else {
          coop_ld(p, S.prod[6 * i + j]);
        }
        if (i + j == k) fp2_add(lo, lo, p);
        else fp2_add(hiacc, hiacc, p);
      }
    }
    // up to six staged products per half (limbs below 6 * 2^28 < 2^31): one carry pass brings the limbs of both halves
    // back under 2^28 before xi doubles one of them; the value (below 20 p) is reduced once, at the end
    fp2_norm(hiacc, hiacc);
    fp2_mul_xi(hiacc, hiacc);
    fp2_norm(lo, lo);
    fp2_add(lo, lo, hiacc);
    fp2_reduce(lo, lo);
    coop_st(dst.c[k], lo);
  }
  __syncthreads();
}

// dst = a * b   (dst may alias a or b: products are staged in S.prod first)
__device__ __noinline__ void coop_mul(coop_shared& S, coop_f12& dst, const coop_f12& a, const coop_f12& b) {
  const int pr = coop_pair();
  for (int r = 0; r < 2; r++) {
    const int q = pr + 32 * r;
    if (q < 36) {
      hfp2 x, y, p;
      coop_ld(x, a.c[q / 6]);
      coop_ld(y, b.c[q % 6]);
      fp2_mul(p, x, y);
      coop_st(S.prod[q], p);
    }
  }
  __syncthreads();
  coop_reduce<0>(S, dst);
}
// the 21 index pairs i <= j in row-major order
__device__ __forceinline__ void coop_sqr_idx(int q, int& i, int& j, int& w) {
  int ii = 0, base = 0;
  while (q >= base + (6 - ii)) {
    base += 6 - ii;
    ii++;
  }
  i = ii;
  j = ii + (q - base);
  w = (i == j) ? 1 : 2;
}
__device__ __noinline__ void coop_sqr(coop_shared& S, coop_f12& dst, const coop_f12& a) {
  const int q = coop_pair();
  if (q < 21) {
    int i, j, w;
    coop_sqr_idx(q, i, j, w);
    hfp2 x, y, p;
    coop_ld(x, a.c[i]);
    coop_ld(y, a.c[j]);
    fp2_mul(p, x, y);
    coop_st(S.prod[q], p);
  }
  __syncthreads();
  coop_reduce<1>(S, dst);
}
// f *= (l0 + l2 w^2 + l3 w^3), the line in S.line[set][0..2]
__device__ __noinline__ void coop_mul_line(coop_shared& S, coop_f12& f, int set) {
  const int q = coop_pair();
  if (q < 18) {
    hfp2 x, y, p;
    coop_ld(x, f.c[q / 3]);
    coop_ld(y, S.line[set][q % 3]);
    fp2_mul(p, x, y);
    coop_st(S.prod[q], p);
  }
  __syncthreads();
  coop_reduce<2>(S, f);
}
__device__ __forceinline__ void coop_copy(coop_f12& dst, const coop_f12& a) {
  for (int w = threadIdx.x; w < 6 * COOP_FP2_WORDS; w += BLS_BLOCK) (&dst.c[0][0])[w] = (&a.c[0][0])[w];
  __syncthreads();
}
// a^(p^6): negate the odd coefficients
__device__ __forceinline__ void coop_conj(coop_f12& dst, const coop_f12& a) {
  const int k = coop_pair();
  if (k < 6) {
    hfp2 x;
    coop_ld(x, a.c[k]);
    if (k & 1) fp2_neg(x, x);
    coop_st(dst.c[k], x);
  }
  __syncthreads();
}
// a^(p^J): coefficient k -> conj^J(c_k) * FROBJ[k]
template <int J>
__device__ __forceinline__ void coop_frob(coop_f12& dst, const coop_f12& a) {
  const int k = coop_pair();
  if (k < 6) {
    hfp2 x, c;
    coop_ld(x, a.c[k]);
    if (J == 1) fp2_conj(c, x);
    else c = x;
    if (k) fp2_mul_const(x, c, (J == 1) ? FROB1[k] : FROB2[k]);
    else x = c;
    coop_st(dst.c[k], x);
  }
  __syncthreads();
}
// tower <-> coefficient order on lane pair 0 (for the one inversion of the easy part)
__device__ __forceinline__ void coop_to_tower(fp12_t<hfp2>& r, const coop_f12& a) {
  coop_ld(r.c0.a0, a.c[0]);
  coop_ld(r.c1.a0, a.c[1]);
  coop_ld(r.c0.a1, a.c[2]);
  coop_ld(r.c1.a1, a.c[3]);
  coop_ld(r.c0.a2, a.c[4]);
  coop_ld(r.c1.a2, a.c[5]);
}
__device__ __forceinline__ void coop_from_tower(coop_f12& d, const fp12_t<hfp2>& r) {
  coop_st(d.c[0], r.c0.a0);
  coop_st(d.c[1], r.c1.a0);
  coop_st(d.c[2], r.c0.a1);
  coop_st(d.c[3], r.c1.a1);
  coop_st(d.c[4], r.c0.a2);
  coop_st(d.c[5], r.c1.a2);
}
// dst = a^2 for a in the cyclotomic subgroup (Granger-Scott, the formulas of fp12_cyclotomic_sqr_body in this basis: the
// Fp4 pairs are (c0, c3), (c1, c4), (c2, c5)).  Nine Fp2 SQUARINGS in one round -- a^2, b^2, (a + b)^2 of each pair, one
// multiplication per lane instead of the fused two-product pass of a general Fp2 product -- then six lane pairs combine
// three staged squares each with their own old coefficient (3 t -+ 2 z) and reduce once.  dst may alias a.
__device__ __noinline__ void coop_cyc_sqr(coop_shared& S, coop_f12& dst, const coop_f12& a) {
  const int q = coop_pair();
  if (q < 9) {
    const int m = q / 3, s = q % 3;
    hfp2 x, y, p;
    coop_ld(x, a.c[m]);
    coop_ld(y, a.c[m + 3]);
    if (s == 1) {
      x = y;
    } else if (s == 2) {
      fp2_add(x, x, y);
      fp2_norm(x, x);
    }
    fp2_sqr(p, x);
    coop_st(S.prod[q], p);
  }
  __syncthreads();
  if (q < 6) {
    // coefficient k takes the even half (A + xi B) or the odd half (C - A - B) of the Fp4 square m:
    //   c0 <- even(0), c3 <- odd(0);  c2 <- even(1), c5 <- odd(1);  c4 <- even(2), c1 <- xi * odd(2)
    const int k = q, m = (k == 0 || k == 3) ? 0 : (k == 2 || k == 5) ? 1 : 2;
    hfp2 A, B, t, z, r;
    coop_ld(A, S.prod[3 * m]);
    coop_ld(B, S.prod[3 * m + 1]);
    if (k & 1) {
      hfp2 C;
      coop_ld(C, S.prod[3 * m + 2]);
      fp2_sub(t, C, A);
      fp2_sub(t, t, B);
      fp2_norm(t, t);
      if (k == 1) {
        fp2_mul_xi(t, t);
        fp2_norm(t, t);
      }
    } else {
      fp2_mul_xi(t, B);
      fp2_add(t, t, A);
      fp2_norm(t, t);
    }
    coop_ld(z, a.c[k]);
    if (k & 1) fp2_add(r, t, z);
    else fp2_sub(r, t, z);
    fp2_dbl(r, r);
    fp2_add(r, r, t);
    fp2_reduce(r, r);
    coop_st(dst.c[k], r);
  }
  __syncthreads();
}
// dst = a^x (x < 0) for a in the cyclotomic subgroup: one round per squaring
__device__ __noinline__ void coop_pow_x(coop_shared& S, coop_f12& dst, const coop_f12& a) {
  coop_copy(S.acc, a);
  for (int i = 62; i >= 0; i--) {
    coop_cyc_sqr(S, S.acc, S.acc);
    if ((BLS_X_ABS >> i) & 1) coop_mul(S, S.acc, S.acc, a);
  }
  coop_conj(dst, S.acc);
}

// Up to six independent Fp2 products per point step and pair: the owners (lane pair k for pair k) stage operand pairs in
// S.job[k][j], lane pair 6 k + j multiplies, the owners read S.res[k][j].  Every lane pair runs the same product code.
__device__ __forceinline__ void coop_jobs(coop_shared& S, int njobs) {
  __syncthreads();
  const int me = coop_pair();
  if (me < 12) {
    const int k = me / 6, j = me % 6;
    if (j < njobs) {
      hfp2 x, y, p;
      coop_ld(x, S.job[k][j][0]);
      coop_ld(y, S.job[k][j][1]);
      fp2_mul(p, x, y);
      coop_st(S.res[k][j], p);
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void coop_job_put(coop_shared& S, int k, int j, const hfp2& a, const hfp2& b) {
  coop_st(S.job[k][j][0], a);
  coop_st(S.job[k][j][1], b);
}

// f <- f^2 with the FIRST round of point-step products riding along: lane pairs 0..20 multiply the 21 coefficient pairs of
// the squaring, lane pairs 21..30 the ten staged jobs S.job[k][0..4] (the same code on every lane pair: two LDS operands,
// one product, one LDS result), so that a Miller iteration needs one product round less.
__device__ __noinline__ void coop_sqr_with_jobs(coop_shared& S, coop_f12& f) {
  __syncthreads();                 // the owners' job operands are staged
  const int q = coop_pair();
  const uint32_t *pa, *pb;
  uint32_t* pd;
  if (q < 21) {
    int i, j, w;
    coop_sqr_idx(q, i, j, w);
    pa = f.c[i];
    pb = f.c[j];
    pd = S.prod[q];
  } else {
    const int jj = q < 31 ? q - 21 : 0, k = jj / 5, j = jj % 5;
    pa = S.job[k][j][0];
    pb = S.job[k][j][1];
    pd = S.res[k][j];
  }
  if (q < 31) {
    hfp2 x, y, p;
    coop_ld(x, pa);
    coop_ld(y, pb);
    fp2_mul(p, x, y);
    coop_st(pd, p);
  }
  __syncthreads();
  coop_reduce<1>(S, f);
}

// f *= line `set` with the SECOND round of point-step products riding along (lane pairs 0..17: the 18 products of the
// sparse multiplication; lane pairs 18..29: the staged jobs S.job[k][0..5]).  Used when pair 1's line comes from the -g2
// table: that line is complete after the first round, so its multiplication need not wait for the general pair's step.
__device__ __noinline__ void coop_mul_line_with_jobs(coop_shared& S, coop_f12& f, int set) {
  __syncthreads();                 // line `set` and the owners' job operands are staged
  const int q = coop_pair();
  const uint32_t *pa, *pb;
  uint32_t* pd;
  if (q < 18) {
    pa = f.c[q / 3];
    pb = S.line[set][q % 3];
    pd = S.prod[q];
  } else {
    const int jj = q < 30 ? q - 18 : 0, k = jj / 6, j = jj % 6;
    pa = S.job[k][j][0];
    pb = S.job[k][j][1];
    pd = S.res[k][j];
  }
  if (q < 30) {
    hfp2 x, y, p;
    coop_ld(x, pa);
    coop_ld(y, pb);
    fp2_mul(p, x, y);
    coop_st(pd, p);
  }
  __syncthreads();
  coop_reduce<2>(S, f);
}

// Miller loop of two pairs into S.f.  fixed_g2: pair 1's G2 member is a constant with a line table -- 1: -g2, 2: -[c] g2 (the
// pair that balances an uncleared message point, csrc/g2neg_lines.cuh) --, 0: both pairs are general.
// Lane pair k (k = 0, 1) owns pair k's point T.  The doubling step -- nine dependent Fp2 products when one lane pair runs
// it alone, the serial part of the loop -- is cut into two rounds of independent products spread over lane pairs
// (coop_jobs: XY, Y^2, Z^2, X^2, (Y+Z)^2, then XY(B-F), (B+F)^2, E^2, BH and the two line scalings), with the owner doing
// the limb-wise glue in between exactly as miller_dbl_step does (pairing.cuh; same bounds).  The five addition steps of
// the loop stay on the owners.
__device__ __noinline__ void coop_miller2(coop_shared& S, const g1_aff* P, const aff<hfp2>* Q, int fixed_g2) {
  const int me = coop_pair();
  const bool second = me == 1;
  const bool owner = me < 2;
  const bool table = second && fixed_g2;          // this owner scales rows of the -g2 line table instead of stepping a point
  const uint32_t (*LINES)[6 * FP_NL] = fixed_g2 == 2 ? G2NEGC_LINES : G2NEG_LINES;   // 2: the table of -[c] g2 (uncleared message points)
  g1_aff Pm;
  aff<hfp2> Qm;
  fp_sel(Pm.x, second, P[1].x, P[0].x);
  fp_sel(Pm.y, second, P[1].y, P[0].y);
  fp_sel(Qm.x.v, second, Q[1].x.v, Q[0].x.v);
  fp_sel(Qm.y.v, second, Q[1].y.v, Q[0].y.v);
  hfp2 xp2, yp2;                                 // (xP, 0), (yP, 0): Fp scalings as Fp2 products, so that all jobs are alike
  {
    fp z;
    fp_zero(z);
    fp_sel(xp2.v, lane_hi(), z, Pm.x);
    fp_sel(yp2.v, lane_hi(), z, Pm.y);
  }
  g2_hom_t<hfp2> T;
  T.x = Qm.x;
  T.y = Qm.y;
  fp2_one(T.z);
  {  // f = 1
    if (me < 6) {
      hfp2 x;
      if (me == 0) fp2_one(x);
      else fp2_zero(x);
      coop_st(S.f.c[me], x);
    }
    __syncthreads();
  }
  hfp2 l0, l2, l3, t;
  int row = 0;
  for (int i = 62; i >= 0; i--) {
    // ---- doubling step (its first product round shares a round with the squaring of f)
    hfp2 a, b, c, e, f, h, g, s;
    if (owner) {
      if (table) {
        fp2_load(l0, &LINES[row][0]);
        fp2_load(t, &LINES[row][2 * FP_NL]);
        coop_job_put(S, me, 0, t, xp2);
        fp2_load(t, &LINES[row][4 * FP_NL]);
        coop_job_put(S, me, 1, t, yp2);
      } else {
        fp2_add(h, T.y, T.z);
        fp2_norm(h, h);
        coop_job_put(S, me, 0, T.x, T.y);
        coop_job_put(S, me, 1, T.y, T.y);
        coop_job_put(S, me, 2, T.z, T.z);
        coop_job_put(S, me, 3, T.x, T.x);
        coop_job_put(S, me, 4, h, h);
      }
    }
    if (i != 62) coop_sqr_with_jobs(S, S.f);
    else coop_jobs(S, 5);
    if (owner) {
      if (table) {
        coop_ld(l2, S.res[me][0]);
        coop_ld(l3, S.res[me][1]);
      } else {
        coop_ld(a, S.res[me][0]);       // XY
        coop_ld(b, S.res[me][1]);       // B = Y^2
        coop_ld(c, S.res[me][2]);       // C = Z^2
        coop_ld(s, S.res[me][3]);       // X^2
        coop_ld(h, S.res[me][4]);       // (Y + Z)^2
        fp2_mul_xi(e, c);
        fp2_dbl(g, e);
        fp2_add(e, g, e);
        fp2_reduce(e, e);
        fp2_dbl(e, e);
        fp2_dbl(e, e);
        fp2_norm(e, e);                 // E = 3b'C
        fp2_dbl(f, e);
        fp2_add(f, f, e);               // F = 3E
        fp2_sub(h, h, b);
        fp2_sub(h, h, c);
        fp2_norm(h, h);                 // H = 2YZ
        fp2_sub(l0, b, e);
        fp2_norm(l0, l0);
        fp2_dbl(g, s);
        fp2_add(g, g, s);
        fp2_neg(g, g);
        fp2_norm(g, g);                 // -3X^2
        fp2_sub(t, b, f);
        fp2_norm(t, t);
        coop_job_put(S, me, 0, a, t);   // XY (B - F)
        fp2_add(t, b, f);
        fp2_norm(t, t);
        coop_job_put(S, me, 1, t, t);   // (B + F)^2
        coop_job_put(S, me, 2, e, e);   // E^2
        coop_job_put(S, me, 3, b, h);   // BH
        coop_job_put(S, me, 4, g, xp2); // l2
        coop_job_put(S, me, 5, h, yp2); // l3
      }
      if (table) {                      // the table pair's line is complete: multiply it in while the other pair steps
        coop_st(S.line[1][0], l0);
        coop_st(S.line[1][1], l2);
        coop_st(S.line[1][2], l3);
      }
    }
    if (fixed_g2) coop_mul_line_with_jobs(S, S.f, 1);
    else coop_jobs(S, 6);
    if (owner) {
      if (!table) {
        coop_ld(g, S.res[me][0]);
        fp2_dbl(g, g);
        fp2_reduce(T.x, g);             // X3 = 2XY(B - F)
        coop_ld(g, S.res[me][1]);
        coop_ld(s, S.res[me][2]);
        fp2_dbl(a, s);
        fp2_add(a, a, s);
        fp2_norm(a, a);
        fp2_dbl(a, a);
        fp2_dbl(a, a);                  // 12E^2
        fp2_sub(g, g, a);
        fp2_reduce(T.y, g);             // Y3 = (B + F)^2 - 12E^2
        coop_ld(g, S.res[me][3]);
        fp2_dbl(g, g);
        fp2_dbl(g, g);
        fp2_reduce(T.z, g);             // Z3 = 4BH
        coop_ld(l2, S.res[me][4]);
        coop_ld(l3, S.res[me][5]);
      }
      if (!table) {
        coop_st(S.line[me][0], l0);
        coop_st(S.line[me][1], l2);
        coop_st(S.line[me][2], l3);
      }
    }
    __syncthreads();
    coop_mul_line(S, S.f, 0);
    if (!fixed_g2) coop_mul_line(S, S.f, 1);
    row++;
    // ---- addition step (set bits of |x|)
    if ((BLS_X_ABS >> i) & 1) {
      if (owner) {
        if (table) {
          fp2_load(l0, &LINES[row][0]);
          fp2_load(t, &LINES[row][2 * FP_NL]);
          fp2_mul_fp(l2, t, Pm.x);
          fp2_load(t, &LINES[row][4 * FP_NL]);
          fp2_mul_fp(l3, t, Pm.y);
        } else {
          miller_add_step(T, l0, l2, l3, Qm.x, Qm.y, Pm.x, Pm.y);
        }
        coop_st(S.line[me][0], l0);
        coop_st(S.line[me][1], l2);
        coop_st(S.line[me][2], l3);
      }
      __syncthreads();
      coop_mul_line(S, S.f, 0);
      coop_mul_line(S, S.f, 1);
      row++;
    }
  }
  coop_conj(S.f, S.f);
}

// easy part of the final exponentiation: S.f <- S.f^((p^6 - 1)(p^2 + 1))
__device__ __noinline__ void coop_final_easy(coop_shared& S) {
  if (coop_pair() == 0) {
    fp12_t<hfp2> a, b;
    coop_to_tower(a, S.f);
    fp12_inv(b, a);
    coop_from_tower(S.t, b);
  }
  __syncthreads();
  coop_conj(S.u, S.f);
  coop_mul(S, S.f, S.u, S.t);
  coop_frob<2>(S.t, S.f);
  coop_mul(S, S.f, S.t, S.f);
}
// final exponentiation of S.f (as pairing.cuh final_exponentiation) and comparison with 1; returns the status code
__device__ __noinline__ int coop_final_verdict(coop_shared& S) {
  coop_final_easy(S);
  // hard part: t = f^((x-1)^2 (x+p) (x^2+p^2-1)) * f^3
  coop_pow_x(S, S.t, S.f);
  coop_conj(S.u, S.f);
  coop_mul(S, S.t, S.t, S.u);      // f^(x-1)
  coop_pow_x(S, S.u, S.t);
  coop_conj(S.v, S.t);
  coop_mul(S, S.t, S.u, S.v);      // f^((x-1)^2)
  coop_pow_x(S, S.u, S.t);
  coop_frob<1>(S.v, S.t);
  coop_mul(S, S.t, S.u, S.v);      // ^(x+p)
  coop_pow_x(S, S.u, S.t);
  coop_pow_x(S, S.u, S.u);         // t^(x^2)
  coop_frob<2>(S.v, S.t);
  coop_mul(S, S.u, S.u, S.v);
  coop_conj(S.v, S.t);
  coop_mul(S, S.t, S.u, S.v);      // ^(x^2+p^2-1)
  coop_sqr(S, S.u, S.f);
  coop_mul(S, S.u, S.u, S.f);      // f^3
  coop_mul(S, S.t, S.t, S.u);
  // == 1 ?
  if (threadIdx.x == 0) S.flag = 1;
  __syncthreads();
  const int k = coop_pair();
  if (k < 6) {
    hfp2 x, one;
    coop_ld(x, S.t.c[k]);
    bool ok;
    if (k == 0) {
      fp2_one(one);
      ok = fp2_eq(x, one);
    } else {
      ok = fp2_is_zero(x);
    }
    if (!ok) S.flag = 0;
  }
  __syncthreads();
  return S.flag ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}
