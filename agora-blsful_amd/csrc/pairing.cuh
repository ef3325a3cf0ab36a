// Optimal-ate Miller loop and final exponentiation for BLS12-381 (x = -0xd201000000010000).
// Replaces `multi_miller_loop(..).final_exponentiation()` + `Gt::is_identity` of the un-vendored backend
// (reference call sites src/helpers.rs:50,62 and src/traits/sig_core.rs:138-141,173).  Only the verdict
// "product of pairings == 1" is consumed by the reference's verify path, so line values are scaled freely by
// proper-subfield factors (they vanish in the final exponentiation) and the final exponentiation computes the
// cube of the canonical value (gcd(3, r) = 1: same verdict).
#pragma once
#include "curve.cuh"
#include "g2neg_lines.cuh"

// BLS_INLINE_MILLER = 1 inlines the Miller steps and the LDS-accumulator operations into their kernel (only the multiplier
// leaves stay calls): nothing then travels by reference through scratch between them
#ifndef BLS_INLINE_MILLER
#define BLS_INLINE_MILLER 0
#endif
#if BLS_INLINE_MILLER
#define BLS_STEP_FN BLS_FN
#else
#define BLS_STEP_FN BLS_NOINLINE
#endif

// homogeneous projective point on the twist, used only inside the Miller loop
template <class F2>
struct g2_hom_t {
  F2 x, y, z;
};
typedef g2_hom_t<fp2> g2_hom;

// Bounds (fp.cuh): T enters and leaves the steps reduced; the line coefficients leave normalised (what
// fp12_mul_by_line expects).  The small-constant multiples (3b' = 12 (1 + u), 3, 4, 12) are limb-wise additions, so
// the steps fp_norm / fp_reduce where a chain would pass 2^31 or feed a product with more than 2^29 per limb.

// Where a step finds and leaves its state: the steps below read T, P (and Q for additions) through an accessor, so that the same
// arithmetic runs on values held in registers (miller_regs: every caller but one) or in LDS (kernels.cuh k_lines2s, whose step
// functions take no operand by reference).  Every read happens where the value is used and every write as soon as it is known.
template <class F2>
struct miller_regs {
  g2_hom_t<F2>& t;
  const fp& xp_;
  const fp& yp_;
  const F2* xq_;
  const F2* yq_;
  BLS_MFN void ld_tx(F2& r) const { r = t.x; }
  BLS_MFN void ld_ty(F2& r) const { r = t.y; }
  BLS_MFN void ld_tz(F2& r) const { r = t.z; }
  BLS_MFN void st_tx(const F2& v) const { t.x = v; }
  BLS_MFN void st_ty(const F2& v) const { t.y = v; }
  BLS_MFN void st_tz(const F2& v) const { t.z = v; }
  BLS_MFN void ld_xp(fp& r) const { r = xp_; }
  BLS_MFN void ld_yp(fp& r) const { r = yp_; }
  BLS_MFN void ld_xq(F2& r) const { r = *xq_; }
  BLS_MFN void ld_yq(F2& r) const { r = *yq_; }
};

// T <- 2T and the tangent line at T evaluated at P = (xp, yp):
//   l0 = Y^2 - 3b'Z^2,  l2 = -3X^2 xp,  l3 = 2YZ yp     (coefficients of w^0, w^2, w^3)
template <class F2, class ST>
BLS_FN void miller_dbl_step_at(const ST& st, F2& l0, F2& l2, F2& l3) {
  F2 x, y, z, a, b, c, e, f, h, g, s;
  fp k;
  st.ld_tx(x);
  st.ld_ty(y);
  st.ld_tz(z);
  fp2_mul(a, x, y);      // XY
  fp2_sqr(b, y);         // B = Y^2
  fp2_sqr(c, z);         // C = Z^2
  fp2_mul_xi(e, c);
  fp2_dbl(g, e);
  fp2_add(e, g, e);      // 3 (1 + u) C
  fp2_reduce(e, e);
  fp2_dbl(e, e);
  fp2_dbl(e, e);
  fp2_norm(e, e);        // E = 3b'C = 12 (1 + u) C
  fp2_dbl(f, e);
  fp2_add(f, f, e);      // F = 3E
  fp2_add(h, y, z);
  fp2_norm(h, h);
  fp2_sqr(h, h);
  fp2_sub(h, h, b);
  fp2_sub(h, h, c);      // H = 2YZ
  fp2_norm(h, h);
  fp2_sqr(s, x);         // X^2
  // line
  fp2_sub(l0, b, e);
  fp2_norm(l0, l0);
  fp2_dbl(g, s);
  fp2_add(g, g, s);      // 3X^2
  fp2_neg(g, g);
  st.ld_xp(k);
  fp2_mul_fp(l2, g, k);
  st.ld_yp(k);
  fp2_mul_fp(l3, h, k);
  // point: X3 = 2XY(B - F), Y3 = (B + F)^2 - 12E^2, Z3 = 4BH
  fp2_sub(g, b, f);
  fp2_norm(g, g);
  fp2_mul(g, a, g);
  fp2_dbl(g, g);
  fp2_reduce(g, g);
  st.st_tx(g);
  fp2_add(g, b, f);
  fp2_norm(g, g);
  fp2_sqr(g, g);
  fp2_sqr(s, e);
  fp2_dbl(a, s);
  fp2_add(a, a, s);      // 3E^2
  fp2_norm(a, a);
  fp2_dbl(a, a);
  fp2_dbl(a, a);         // 12E^2
  fp2_sub(g, g, a);
  fp2_reduce(g, g);
  st.st_ty(g);
  fp2_mul(g, b, h);
  fp2_dbl(g, g);
  fp2_dbl(g, g);
  fp2_reduce(g, g);
  st.st_tz(g);
}
template <class F2>
BLS_STEP_FN void miller_dbl_step(g2_hom_t<F2>& t, F2& l0, F2& l2, F2& l3, const fp& xp, const fp& yp) {
  const miller_regs<F2> st = {t, xp, yp, nullptr, nullptr};
  miller_dbl_step_at(st, l0, l2, l3);
}

// T <- T + Q and the chord line through T and Q evaluated at P:
//   l0 = theta xq - lambda yq,  l2 = -theta xp,  l3 = lambda yp
template <class F2, class ST>
BLS_FN void miller_add_step_at(const ST& st, F2& l0, F2& l2, F2& l3) {
  F2 th, la, c, d, e, f, g, h, s, q, z, u;
  fp k;
  st.ld_yq(q);
  st.ld_tz(z);
  fp2_mul(th, q, z);
  st.ld_ty(u);
  fp2_sub(th, u, th);    // theta = Y - yq Z
  st.ld_xq(q);
  fp2_mul(la, q, z);
  st.ld_tx(u);
  fp2_sub(la, u, la);    // lambda = X - xq Z
  fp2_norm(th, th);
  fp2_norm(la, la);
  fp2_mul(l0, th, q);
  st.ld_yq(q);
  fp2_mul(s, la, q);
  fp2_sub(l0, l0, s);
  fp2_norm(l0, l0);
  fp2_neg(s, th);
  st.ld_xp(k);
  fp2_mul_fp(l2, s, k);
  st.ld_yp(k);
  fp2_mul_fp(l3, la, k);
  fp2_sqr(c, th);
  fp2_sqr(d, la);
  fp2_mul(e, la, d);
  st.ld_tz(z);
  fp2_mul(f, z, c);
  st.ld_tx(u);
  fp2_mul(g, u, d);
  fp2_add(h, e, f);
  fp2_sub(h, h, g);
  fp2_sub(h, h, g);
  fp2_norm(h, h);
  fp2_mul(u, la, h);
  st.st_tx(u);
  fp2_sub(s, g, h);
  fp2_mul(s, th, s);
  st.ld_ty(u);
  fp2_mul(g, e, u);
  fp2_sub(s, s, g);
  fp2_reduce(s, s);
  st.st_ty(s);
  st.ld_tz(z);
  fp2_mul(z, z, e);
  st.st_tz(z);
}
template <class F2>
BLS_STEP_FN void miller_add_step(g2_hom_t<F2>& t, F2& l0, F2& l2, F2& l3, const F2& xq, const F2& yq, const fp& xp,
                                  const fp& yp) {
  const miller_regs<F2> st = {t, xp, yp, &xq, &yq};
  miller_add_step_at(st, l0, l2, l3);
}

// The Miller accumulator is addressed through acc_* so that the same loops run with f in private memory (an fp12_t, below)
// or in LDS (the lane-split kernels' f12_sh handle, tower_split.cuh).
template <class F2>
BLS_FN void acc_one(fp12_t<F2>& f) { fp12_one(f); }
template <class F2>
BLS_FN void acc_sqr(fp12_t<F2>& f) { fp12_sqr(f, f); }
template <class F2>
BLS_FN void acc_mul_line(fp12_t<F2>& f, const F2& l0, const F2& l2, const F2& l3) { fp12_mul_by_line(f, l0, l2, l3); }
// Two line values of one step.  BLS_MERGE_LINES = 1 multiplies them together first (23 instead of 26 Fp2 products,
// tower.cuh fp12_mul_by_2lines); measured on MI355X the merged form saves 6 % of the instructions of the Miller kernel but
// needs ~250 live registers, and its spills doubled the kernel's scratch traffic for no gain in time, so the default
// keeps two sparse multiplications.
#ifndef BLS_MERGE_LINES
#define BLS_MERGE_LINES 0
#endif
template <class F2>
BLS_FN void acc_mul_2lines(fp12_t<F2>& f, const F2& a0, const F2& a2, const F2& a3, const F2& b0, const F2& b2, const F2& b3) {
#if BLS_MERGE_LINES
  fp12_mul_by_2lines(f, a0, a2, a3, b0, b2, b3);
#else
  fp12_mul_by_line(f, a0, a2, a3);
  fp12_mul_by_line(f, b0, b2, b3);
#endif
}
template <class F2>
BLS_FN void acc_finish(fp12_t<F2>& f) { fp12_conj(f, f); }   // x < 0

// f <- conj( prod_i f_{|x|,Q_i}(P_i) ).  Pairs with a point at infinity contribute 1.
template <int N, class ACC, class F2>
BLS_FN void miller_loop(ACC& f, const g1_aff* P, const aff<F2>* Q) {
  g2_hom_t<F2> T[N];
  bool skip[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    skip[k] = P[k].inf || Q[k].inf;
    T[k].x = Q[k].x;
    T[k].y = Q[k].y;
    fp2_one(T[k].z);
  }
  acc_one(f);
  F2 l0, l2, l3, m0, m2, m3;
  const bool both = (N == 2) && !skip[0] && !skip[N - 1];
  for (int i = 62; i >= 0; i--) {
    if (i != 62) acc_sqr(f);
    for (int step = 0; step < 2; step++) {          // 0: doubling, 1: addition (only at set bits of |x|)
      if (step == 1 && !((BLS_X_ABS >> i) & 1)) break;
      if (both) {                                     // two pairs: merge their line values before touching f
        if (step == 0) {
          miller_dbl_step(T[0], l0, l2, l3, P[0].x, P[0].y);
          miller_dbl_step(T[N - 1], m0, m2, m3, P[N - 1].x, P[N - 1].y);
        } else {
          miller_add_step(T[0], l0, l2, l3, Q[0].x, Q[0].y, P[0].x, P[0].y);
          miller_add_step(T[N - 1], m0, m2, m3, Q[N - 1].x, Q[N - 1].y, P[N - 1].x, P[N - 1].y);
        }
        acc_mul_2lines(f, l0, l2, l3, m0, m2, m3);
      } else {
#pragma unroll
        for (int k = 0; k < N; k++) {
          if (!skip[k]) {
            if (step == 0) miller_dbl_step(T[k], l0, l2, l3, P[k].x, P[k].y);
            else miller_add_step(T[k], l0, l2, l3, Q[k].x, Q[k].y, P[k].x, P[k].y);
            acc_mul_line(f, l0, l2, l3);
          }
        }
      }
    }
  }
  acc_finish(f);
}

// The same product for exactly two pairs whose SECOND G2 argument is the constant -g2 (core_verify of Bls12381G1Impl:
// e(H(m), pk) e(sig, -g2), reference src/traits/sig_core.rs:136-138): that pair's point arithmetic is replaced by the
// precomputed table (G2NEG_LINES, or G2NEGC_LINES for -[c] g2: tools/gen_g2_lines.py); only the two Fp2-by-Fp scalings by (xP, yP) remain.
template <class ACC, class F2>
BLS_FN void miller_loop_fixed_g2(ACC& f, const g1_aff& P0, const aff<F2>& Q0, const g1_aff& P1, const uint32_t (*lines)[6 * FP_NL] = G2NEG_LINES) {
  g2_hom_t<F2> T;
  T.x = Q0.x;
  T.y = Q0.y;
  fp2_one(T.z);
  acc_one(f);
  F2 l0, l2, l3, m0, m2, m3, t;
  int row = 0;
  for (int i = 62; i >= 0; i--) {
    if (i != 62) acc_sqr(f);
    for (int step = 0; step < 2; step++) {          // 0: doubling, 1: addition (only at set bits of |x|)
      if (step == 1 && !((BLS_X_ABS >> i) & 1)) break;
      if (step == 0) miller_dbl_step(T, l0, l2, l3, P0.x, P0.y);
      else miller_add_step(T, l0, l2, l3, Q0.x, Q0.y, P0.x, P0.y);
      fp2_load(m0, &lines[row][0]);
      fp2_load(t, &lines[row][2 * FP_NL]);
      fp2_mul_fp(m2, t, P1.x);
      fp2_load(t, &lines[row][4 * FP_NL]);
      fp2_mul_fp(m3, t, P1.y);
      acc_mul_2lines(f, l0, l2, l3, m0, m2, m3);
      row++;
    }
  }
  acc_finish(f);
}

// ---- the same product with the two line values of a step MERGED where they are computed (round 3) ------------------------
// One step of that loop for a fixed second G2 argument given as a normalised table (g2neg_lines.cuh *_LINES_N: rows (l0/h, g/h),
// the line value is n0 + (n2 x1) w^2 + y1 w^3 with P1 = (x1, y1)): walk T (doubling, or addition of Q0 when add is set), evaluate
// the chord / tangent at P0 and multiply the two sparse values into five coefficients (tower.cuh lines_merge_y).
template <class F2, class ST>
BLS_FN void miller_line5_fixed_at(line5_t<F2>& L, const ST& st, bool add, const fp& x1, const fp& y1, const uint32_t* row) {
  F2 l0, l2, l3, n0, n2, t;
  if (add) miller_add_step_at(st, l0, l2, l3);
  else miller_dbl_step_at(st, l0, l2, l3);
  fp2_load(n0, row);
  fp2_load(t, row + 2 * FP_NL);
  fp2_mul_fp(n2, t, x1);
  lines_merge_y(L, l0, l2, l3, n0, n2, y1);
}
template <class F2>
BLS_FN void miller_line5_fixed(line5_t<F2>& L, g2_hom_t<F2>& T, bool add, const F2& xq, const F2& yq, const fp& xp, const fp& yp, const fp& x1, const fp& y1,
                               const uint32_t* row) {
  const miller_regs<F2> st = {T, xp, yp, &xq, &yq};
  miller_line5_fixed_at(L, st, add, x1, y1, row);
}
// ... and for two general pairs
template <class F2>
BLS_FN void miller_line5_pair(line5_t<F2>& L, g2_hom_t<F2>& T0, g2_hom_t<F2>& T1, bool add, const aff<F2>* Q, const g1_aff* P) {
  F2 l0, l2, l3, m0, m2, m3;
  if (add) {
    miller_add_step(T0, l0, l2, l3, Q[0].x, Q[0].y, P[0].x, P[0].y);
    miller_add_step(T1, m0, m2, m3, Q[1].x, Q[1].y, P[1].x, P[1].y);
  } else {
    miller_dbl_step(T0, l0, l2, l3, P[0].x, P[0].y);
    miller_dbl_step(T1, m0, m2, m3, P[1].x, P[1].y);
  }
  lines_merge(L, l0, l2, l3, m0, m2, m3);
}
template <class F2>
BLS_FN void acc_set_line5(fp12_t<F2>& f, const line5_t<F2>& L) { fp12_from_line5(f, L); }
template <class F2>
BLS_FN void acc_mul_line5(fp12_t<F2>& f, const line5_t<F2>& L) { fp12_mul_by_line5_body(f, L); }
// The 68 steps in loop order: entry e is an addition step when bit e of this mask is set (the set bits of |x| below the top one
// are 62, 60, 57, 48, 16: additions follow the doublings of those iterations)
#define MILLER_ENTRIES 68
BLS_FN bool miller_entry_is_add(int e) {
  // doubling of iteration i = 62 - d sits at entry d + (number of set bits of |x| among 62 .. i + 1); additions right after
  return e == 1 || e == 4 || e == 8 || e == 18 || e == 51;
}
// f <- conj(f_{|x|,Q0}(P0) f_{|x|,Q1}(P1)) up to a factor in Fp2 (killed by the final exponentiation), Q1 the fixed argument of `rows`
template <class ACC, class F2>
BLS_FN void miller_loop_fixed_g2_merged(ACC& f, const g1_aff& P0, const aff<F2>& Q0, const g1_aff& P1, const uint32_t (*rows)[4 * FP_NL]) {
  g2_hom_t<F2> T;
  T.x = Q0.x;
  T.y = Q0.y;
  fp2_one(T.z);
  line5_t<F2> L;
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    const bool add = miller_entry_is_add(e);
    miller_line5_fixed(L, T, add, Q0.x, Q0.y, P0.x, P0.y, P1.x, P1.y, rows[e]);
    if (e == 0) {
      acc_set_line5(f, L);
    } else {
      if (!add) acc_sqr(f);
      acc_mul_line5(f, L);
    }
  }
  acc_finish(f);
}
template <class ACC, class F2>
BLS_FN void miller_loop2_merged(ACC& f, const g1_aff* P, const aff<F2>* Q) {
  g2_hom_t<F2> T0, T1;
  T0.x = Q[0].x;
  T0.y = Q[0].y;
  fp2_one(T0.z);
  T1.x = Q[1].x;
  T1.y = Q[1].y;
  fp2_one(T1.z);
  line5_t<F2> L;
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    const bool add = miller_entry_is_add(e);
    miller_line5_pair(L, T0, T1, add, Q, P);
    if (e == 0) {
      acc_set_line5(f, L);
    } else {
      if (!add) acc_sqr(f);
      acc_mul_line5(f, L);
    }
  }
  acc_finish(f);
}

// a^x for a in the cyclotomic subgroup (x < 0: conjugate of a^|x|): the plain chain of 63 Granger-Scott squarings and five
// multiplications; fp12_pow_x below prefers the compressed chain
template <class F2>
BLS_NOINLINE void fp12_pow_x_plain(fp12_t<F2>& r, const fp12_t<F2>& a) {
  fp12_t<F2> acc = a;
  for (int i = 62; i >= 0; i--) {
    fp12_cyclotomic_sqr(acc, acc);
    if ((BLS_X_ABS >> i) & 1) fp12_mul(acc, acc, a);
  }
  fp12_conj(r, acc);
}

// ---- a^|x| with Karabina's compressed squarings ----------------------------------------------------------------------
// In the cyclotomic subgroup the Granger-Scott update of the coefficient pairs (z2, z3) and (z4, z5) (notation of
// fp12_cyclotomic_sqr_body) does not involve (z0, z1): a squaring of the COMPRESSED element (z2, z3, z4, z5) is two Fp4
// squarings instead of three.  |x| = 2^63 + 2^62 + 2^60 + 2^57 + 2^48 + 2^16, so a^|x| is the product of six of the 63
// successive squarings; those six are decompressed together,
//     z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2),      z0 = (2 z1^2 + z2 z5 - 3 z3 z4) xi + 1,
// with ONE Fp2 inversion for the six denominators (Montgomery's trick; the inversion itself is the safegcd of fp.cuh).
// Checked against the plain chain in tests/hostsim for both tower instantiations.  Returns false -- the caller then runs the
// plain chain -- when one of the six has z2 = 0 (the other decompression formula; never seen for honest inputs).
template <class F2>
struct cyc_c {
  F2 z2, z3, z4, z5;
};
template <class F2>
BLS_FN void cyc_compress(cyc_c<F2>& c, const fp12_t<F2>& f) {
  c.z2 = f.c1.a0;
  c.z3 = f.c0.a2;
  c.z4 = f.c0.a1;
  c.z5 = f.c1.a2;
}
template <class F2>
BLS_FN void cyc_out(F2& r, const F2& t, const F2& z, bool plus) {   // 3 t +- 2 z, reduced
  F2 u;
  if (plus) fp2_add(u, t, z);
  else fp2_sub(u, t, z);
  fp2_dbl(u, u);
  fp2_add(u, u, t);
  fp2_reduce(r, u);
}
// (the Fp4 squarings leave their outputs as plain sums and every new coordinate 3 t +- 2 z is ONE reduction pass that forms the
// combination on 64 bits, fp_reduce_lin2: no doublings, additions or carry passes in between -- round 3)
template <class F2>
BLS_FN void cyc_c_sqr(cyc_c<F2>& r, const cyc_c<F2>& a) {
  F2 t0, t1, t2, t3, x;
  fp4_sqr_lazy(t0, t1, a.z2, a.z3);
  fp4_sqr_lazy(t2, t3, a.z4, a.z5);
  cyc_c<F2> o;
  fp2_reduce_lin2(o.z4, t0, 3, a.z4, -2);
  fp2_reduce_lin2(o.z5, t1, 3, a.z5, 2);
  fp2_mul_xi(x, t3);
  fp2_reduce_lin2(o.z2, x, 3, a.z2, 2);
  fp2_reduce_lin2(o.z3, t2, 3, a.z3, -2);
  r = o;
}
// f = the cyclotomic element whose compressed form is c, given inv4z2 = 1 / (4 z2)
template <class F2>
BLS_FN void cyc_decompress(fp12_t<F2>& f, const cyc_c<F2>& c, const F2& inv4z2) {
  F2 n, t, z1, z0, one;
  fp2_sqr(n, c.z5);
  fp2_mul_xi(n, n);
  fp2_sqr(t, c.z4);
  fp2_dbl(z1, t);
  fp2_add(t, z1, t);            // 3 z4^2
  fp2_add(n, n, t);
  fp2_dbl(t, c.z3);
  fp2_sub(n, n, t);
  fp2_reduce(n, n);
  fp2_mul(z1, n, inv4z2);
  fp2_sqr(z0, z1);
  fp2_dbl(z0, z0);              // 2 z1^2
  fp2_mul(t, c.z2, c.z5);
  fp2_add(z0, z0, t);
  fp2_mul(t, c.z3, c.z4);
  fp2_dbl(n, t);
  fp2_add(t, n, t);             // 3 z3 z4
  fp2_sub(z0, z0, t);
  fp2_reduce(z0, z0);
  fp2_mul_xi(z0, z0);
  fp2_one(one);
  fp2_add(z0, z0, one);
  fp2_reduce(f.c0.a0, z0);
  f.c1.a1 = z1;
  f.c1.a0 = c.z2;
  f.c0.a2 = c.z3;
  f.c0.a1 = c.z4;
  f.c1.a2 = c.z5;
}
// the six saved squarings -> acc = their product; false if a z2 vanishes
template <class F2>
BLS_NOINLINE bool cyc_product6(fp12_t<F2>& acc, const cyc_c<F2>* s) {
  F2 d[6], pre[6], inv, t;
  bool ok = true;
  for (int i = 0; i < 6; i++) {
    if (fp2_is_zero(s[i].z2)) ok = false;
    fp2_dbl(t, s[i].z2);
    fp2_dbl(t, t);
    fp2_norm(d[i], t);                       // 4 z2
  }
  if (!ok) return false;
  pre[0] = d[0];
  for (int i = 1; i < 6; i++) fp2_mul(pre[i], pre[i - 1], d[i]);
  fp2_inv(inv, pre[5]);
  for (int i = 5; i >= 0; i--) {
    F2 di;
    if (i) {
      fp2_mul(di, inv, pre[i - 1]);          // 1 / d_i
      fp2_mul(inv, inv, d[i]);
    } else {
      di = inv;
    }
    fp12_t<F2> e;
    cyc_decompress(e, s[i], di);
    if (i == 5) acc = e;
    else fp12_mul(acc, acc, e);
  }
  return true;
}
template <class F2>
BLS_NOINLINE bool fp12_pow_x_compressed(fp12_t<F2>& r, const fp12_t<F2>& a) {
  cyc_c<F2> c, s[6];
  {
    fp12_t<F2> ar;
    fp12_reduce(ar, a);                      // a may carry negated limbs (a conjugate)
    cyc_compress(c, ar);
  }
  int k = 0;
  for (int i = 1; i <= 63; i++) {
    cyc_c_sqr(c, c);
    if ((BLS_X_ABS >> i) & 1) s[k++] = c;
  }
  fp12_t<F2> acc;
  if (!cyc_product6(acc, s)) return false;
  fp12_conj(r, acc);
  return true;
}

// The same with a short plain tail (round 4; what k_finalexp2s runs):  |x| = 2^16 + 2^48 + 2^57 (1 + 2^3 + 2^5 + 2^6) and 1 + 8 + 32 + 64 = 105
// = 15 * 7, so with b = a^(2^57)
//     c = b^15 = a^(2^61) conj(b),        b^105 = c^7 = c^8 conj(c)        (the conjugate is the inverse in the cyclotomic subgroup):
// 61 compressed squarings, FOUR powers to decompress (2^16, 2^48, 2^57, 2^61) and four products, plus three Granger-Scott squarings --
// against 63, six and five: two decompressions, six of the fifteen products of the shared inversion and one Fp12 product less per a^x.
#define BLS_XK_LAST 61
#define BLS_XK_SAVE ((1ull << 16) | (1ull << 48) | (1ull << 57) | (1ull << 61))
template <class F2>
BLS_NOINLINE bool fp12_pow_x_compressed4(fp12_t<F2>& r, const fp12_t<F2>& a) {
  cyc_c<F2> c, s[4];
  {
    fp12_t<F2> ar;
    fp12_reduce(ar, a);                      // a may carry negated limbs (a conjugate)
    cyc_compress(c, ar);
  }
  int k = 0;
  for (int i = 1; i <= BLS_XK_LAST; i++) {
    cyc_c_sqr(c, c);
    if ((BLS_XK_SAVE >> i) & 1) s[k++] = c;
  }
  F2 d[4], pre[4], inv, t;
  bool ok = true;
  for (int i = 0; i < 4; i++) {
    if (fp2_is_zero(s[i].z2)) ok = false;
    fp2_dbl(t, s[i].z2);
    fp2_dbl(t, t);
    fp2_norm(d[i], t);                       // 4 z2
  }
  if (!ok) return false;
  pre[0] = d[0];
  for (int i = 1; i < 4; i++) fp2_mul(pre[i], pre[i - 1], d[i]);
  fp2_inv(inv, pre[3]);
  fp12_t<F2> acc, e, cc;
  for (int i = 3; i >= 0; i--) {
    F2 di;
    if (i) {
      fp2_mul(di, inv, pre[i - 1]);          // 1 / d_i
      fp2_mul(inv, inv, d[i]);
    } else {
      di = inv;
    }
    cyc_decompress(e, s[i], di);
    if (i == 3) {
      fp12_reduce(acc, e);                   // a^(2^61)
    } else if (i == 2) {
      fp12_conj(e, e);
      fp12_mul(acc, acc, e);                 // c = b^15
      cc = acc;
      for (int j = 0; j < 3; j++) fp12_cyclotomic_sqr(acc, acc);
      fp12_conj(e, cc);
      fp12_mul(acc, acc, e);                 // c^7 = b^105
    } else {
      fp12_mul(acc, acc, e);
    }
  }
  fp12_conj(r, acc);
  return true;
}

template <class F2>
BLS_FN void fp12_pow_x(fp12_t<F2>& r, const fp12_t<F2>& a) {
  if (!fp12_pow_x_compressed(r, a)) fp12_pow_x_plain(r, a);
}

// f^(3 (p^12 - 1)/r), using 3 (p^4 - p^2 + 1)/r = (x-1)^2 (x+p) (x^2+p^2-1) + 3.
// final_exp_parts: f <- fin after the easy part, t <- f^((x-1)^2 (x+p) (x^2+p^2-1)); the value is t f^3.
template <class F2>
BLS_FN void final_exp_parts(fp12_t<F2>& t, fp12_t<F2>& f, const fp12_t<F2>& fin) {
  fp12_t<F2> u, v;
  // easy part: f^((p^6 - 1)(p^2 + 1))
  fp12_inv(t, fin);
  fp12_conj(f, fin);
  fp12_mul(f, f, t);
  fp12_frob<2>(t, f);
  fp12_mul(f, t, f);
  // hard part
  fp12_pow_x(t, f);
  fp12_conj(u, f);
  fp12_mul(t, t, u);  // f^(x-1)
  fp12_pow_x(u, t);
  fp12_conj(v, t);
  fp12_mul(t, u, v);  // f^((x-1)^2)
  fp12_pow_x(u, t);
  fp12_frob<1>(v, t);
  fp12_mul(t, u, v);  // ^(x+p)
  fp12_pow_x(u, t);
  fp12_pow_x(u, u);   // t^(x^2)
  fp12_frob<2>(v, t);
  fp12_mul(u, u, v);
  fp12_conj(v, t);
  fp12_mul(t, u, v);  // ^(x^2+p^2-1)
}
template <class F2>
BLS_FN void final_exponentiation(fp12_t<F2>& r, const fp12_t<F2>& fin) {
  fp12_t<F2> f, t, u;
  final_exp_parts(t, f, fin);
  fp12_cyclotomic_sqr(u, f);
  fp12_mul(u, u, f);  // f^3
  fp12_mul(r, t, u);
}
// the verdict only: t f^3 == 1  <=>  t f^2 == conj(f)  (f lies in the cyclotomic subgroup after the easy part: its conjugate is its
// inverse) -- one Fp12 product less than forming the value (round 4)
template <class F2>
BLS_FN bool final_exp_is_one(const fp12_t<F2>& fin) {
  fp12_t<F2> f, t, u, c;
  final_exp_parts(t, f, fin);
  fp12_cyclotomic_sqr(u, f);
  fp12_mul(u, t, u);
  fp12_conj(c, f);
  return fp2_eq(u.c0.a0, c.c0.a0) && fp2_eq(u.c0.a1, c.c0.a1) && fp2_eq(u.c0.a2, c.c0.a2) && fp2_eq(u.c1.a0, c.c1.a0) && fp2_eq(u.c1.a1, c.c1.a1) &&
         fp2_eq(u.c1.a2, c.c1.a2);
}
