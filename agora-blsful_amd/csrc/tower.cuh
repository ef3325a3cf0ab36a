// Fp2 / Fp6 / Fp12 tower for BLS12-381:  Fp2 = Fp[u]/(u^2+1),  Fp6 = Fp2[v]/(v^3 - xi),  xi = 1 + u,
// Fp12 = Fp6[w]/(w^2 - v).  An Fp12 element c0 + c1 w holds the w-power coefficients
//   w^0 = c0.a0, w^1 = c1.a0, w^2 = c0.a1, w^3 = c1.a1, w^4 = c0.a2, w^5 = c1.a2.
// Replaces the Fp2/Fp6/Fp12 arithmetic of the reference's un-vendored blst backend
// (call sites: reference src/helpers.rs:44,50,56,62).
#pragma once
#include "fp.cuh"

// ------------------------------------------------------------------ Fp2
struct fp2 {
  fp c0, c1;
};

BLS_FN void fp2_load(fp2& r, const uint32_t* c) {
  fp_load(r.c0, c);
  fp_load(r.c1, c + FP_NL);
}
BLS_FN void fp2_store(uint32_t* c, const fp2& a) {
  fp_store(c, a.c0);
  fp_store(c + FP_NL, a.c1);
}
BLS_FN void fp2_zero(fp2& r) {
  fp_zero(r.c0);
  fp_zero(r.c1);
}
BLS_FN void fp2_one(fp2& r) {
  fp_one(r.c0);
  fp_zero(r.c1);
}
BLS_FN void fp2_reduce_lin2(fp2& r, const fp2& a, int ka, const fp2& b, int kb) {
  fp_reduce_lin2(r.c0, a.c0, ka, b.c0, kb);
  fp_reduce_lin2(r.c1, a.c1, ka, b.c1, kb);
}
BLS_FN void fp2_from_fp(fp2& r, const fp& k) {   // k + 0 u
  r.c0 = k;
  fp_zero(r.c1);
}
BLS_FN bool fp2_is_zero(const fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
BLS_FN bool fp2_eq(const fp2& a, const fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
BLS_FN void fp2_cmov(fp2& r, const fp2& a, bool c) {
  fp_cmov(r.c0, a.c0, c);
  fp_cmov(r.c1, a.c1, c);
}
BLS_FN void fp2_add(fp2& r, const fp2& a, const fp2& b) {
  fp_add(r.c0, a.c0, b.c0);
  fp_add(r.c1, a.c1, b.c1);
}
BLS_FN void fp2_sub(fp2& r, const fp2& a, const fp2& b) {
  fp_sub(r.c0, a.c0, b.c0);
  fp_sub(r.c1, a.c1, b.c1);
}
BLS_FN void fp2_neg(fp2& r, const fp2& a) {
  fp_neg(r.c0, a.c0);
  fp_neg(r.c1, a.c1);
}
BLS_FN void fp2_dbl(fp2& r, const fp2& a) {
  fp_dbl(r.c0, a.c0);
  fp_dbl(r.c1, a.c1);
}
BLS_FN void fp2_conj(fp2& r, const fp2& a) {
  r.c0 = a.c0;
  fp_neg(r.c1, a.c1);
}
// Bound contract of the Fp2 layer (fp.cuh explains the lazy representation): fp2_mul accepts operands whose limbs are
// below 2^29 + a few units ("2N+": the sum of two normalised elements), fp2_sqr a normalised operand ("N+", limbs up to
// 2^28 + a few units: it forms a0 +- a1 itself); both return normalised limbs.
BLS_FN void fp2_norm(fp2& r, const fp2& a) {
  fp_norm(r.c0, a.c0);
  fp_norm(r.c1, a.c1);
}
BLS_FN void fp2_reduce(fp2& r, const fp2& a) {
  fp_reduce(r.c0, a.c0);
  fp_reduce(r.c1, a.c1);
}
// Karatsuba: 3 Fp multiplications
BLS_FN void fp2_mul(fp2& r, const fp2& a, const fp2& b) {
  fp t0, t1, s0, s1, m;
  fp_mul(t0, a.c0, b.c0);
  fp_mul(t1, a.c1, b.c1);
  fp_add(s0, a.c0, a.c1);
  fp_add(s1, b.c0, b.c1);
  fp_norm(s0, s0);
  fp_norm(s1, s1);
  fp_mul(m, s0, s1);
  fp_sub(m, m, t0);
  fp_sub(m, m, t1);
  fp_sub(t0, t0, t1);
  fp_norm(r.c1, m);
  fp_norm(r.c0, t0);
}
// complex squaring: 2 Fp multiplications
BLS_FN void fp2_sqr(fp2& r, const fp2& a) {
  fp s, d, m;
  fp_add(s, a.c0, a.c1);
  fp_sub(d, a.c0, a.c1);
  fp_norm(s, s);
  fp_norm(d, d);
  fp_mul(m, a.c0, a.c1);
  fp_mul(r.c0, s, d);
  fp_dbl(m, m);
  fp_norm(r.c1, m);
}
// a * k for a constant k stored as 2 * FP_NL words (internal form)
BLS_FN void fp2_mul_const(fp2& r, const fp2& a, const uint32_t* k) {
  fp2 g;
  fp2_load(g, k);
  fp2_mul(r, a, g);
}
BLS_FN void fp2_mul_fp(fp2& r, const fp2& a, const fp& k) {
  fp_mul(r.c0, a.c0, k);
  fp_mul(r.c1, a.c1, k);
}
// multiply by xi = 1 + u
BLS_FN void fp2_mul_xi(fp2& r, const fp2& a) {
  fp t;
  fp_sub(t, a.c0, a.c1);
  fp_add(r.c1, a.c0, a.c1);
  r.c0 = t;
}
BLS_NOINLINE void fp2_inv(fp2& r, const fp2& a) {
  fp n, t;
  fp_sqr(n, a.c0);
  fp_sqr(t, a.c1);
  fp_add(n, n, t);
  fp_inv(n, n);
  fp_mul(r.c0, a.c0, n);
  fp_mul(t, a.c1, n);
  fp_neg(r.c1, t);
}
BLS_FN bool fp2_is_square(const fp2& a) {
  fp n, t;
  fp_sqr(n, a.c0);
  fp_sqr(t, a.c1);
  fp_add(n, n, t);
  return fp_is_square(n);
}
// some square root (complex method); false if a is not a square.  Callers fix the sign.
// e / einv (optional): an extra non-zero Fp value whose inverse the caller needs anyway -- it shares the exponentiation
// of the 1/(2s) step (Montgomery's trick), e.g. the denominator of the SSWU map.  einv is written only on success.
BLS_NOINLINE bool fp2_sqrt_inv(fp2& r, const fp2& a, const fp* e, fp* einv) {
  if (fp_is_zero(a.c1)) {
    fp s;
    bool ok;
    fp2 cand;
    if (fp_sqrt(s, a.c0)) {
      cand.c0 = s;
      fp_zero(cand.c1);
      ok = true;
    } else {
      fp na;
      fp_neg(na, a.c0);
      ok = fp_sqrt(s, na);  // sqrt(-1) = u
      fp_zero(cand.c0);
      cand.c1 = s;
    }
    if (ok) {
      r = cand;
      if (e) fp_inv(*einv, *e);
    }
    return ok;
  }
  // a1 != 0.  With n = |a| (the square root of the norm, which exists iff a is a square) and t = (a0 + n) / 2:
  //   s = t^((p+1)/4) satisfies s^2 = t or s^2 = -t (p = 3 mod 4), and a0 = t - a1^2 / (4 t), so
  //   sqrt(a) = s + a1/(2s) u   when s^2 = t,      a1/(2s) + s u   when s^2 = -t.
  // Two exponentiations (sqrt(norm), s) and the inversion 1/(2s) (safegcd, fp.cuh).
  fp n, t, h, s, c2, d;
  fp_sqr(n, a.c0);
  fp_sqr(t, a.c1);
  fp_add(n, n, t);
  if (!fp_sqrt(n, n)) return false;
  fp_load(h, FP_HALF);
  fp_add(t, a.c0, n);
  fp_mul(t, t, h);
  fp_pow(s, t, EXP_PM3D4, EXP_PM3D4_BITS);
  fp_mul(s, s, t);
  fp_sqr(c2, s);
  const bool direct = fp_eq(c2, t);
  fp_dbl(d, s);
  fp_reduce(d, d);
  fp ei;
  if (e) {                  // 1/(2s) and 1/e from one exponentiation
    fp prod, pi;
    fp_mul(prod, d, *e);
    fp_inv(pi, prod);
    fp_mul(ei, pi, d);
    fp_mul(d, pi, *e);
  } else {
    fp_inv(d, d);
  }
  fp_mul(d, a.c1, d);
  fp2 cand;
  cand.c0 = s;
  cand.c1 = d;
  if (!direct) {
    cand.c0 = d;
    cand.c1 = s;
  }
  fp2 chk;
  fp2_sqr(chk, cand);
  if (!fp2_eq(chk, a)) return false;
  r = cand;
  if (e) *einv = ei;
  return true;
}
BLS_FN bool fp2_sqrt(fp2& r, const fp2& a) { return fp2_sqrt_inv(r, a, nullptr, nullptr); }
// RFC 9380 sgn0 for m = 2
BLS_FN uint32_t fp2_sgn0(const fp2& a) {
  fp t0, t1;
  fp_from_mont(t0, a.c0);
  fp_from_mont(t1, a.c1);
  uint32_t s0 = (uint32_t)t0.l[0] & 1u, z0 = fp_is_zero(t0) ? 1u : 0u, s1 = (uint32_t)t1.l[0] & 1u;
  return s0 | (z0 & s1);
}
BLS_FN bool fp2_lex_largest(const fp2& a) {
  if (!fp_is_zero(a.c1)) return fp_lex_largest(a.c1);
  return fp_lex_largest(a.c0);
}

// ------------------------------------------------------------------ Fp6
template <class F2>
struct fp6_t {
  F2 a0, a1, a2;
};
typedef fp6_t<fp2> fp6;

template <class F2>
BLS_FN void fp6_zero(fp6_t<F2>& r) {
  fp2_zero(r.a0);
  fp2_zero(r.a1);
  fp2_zero(r.a2);
}
template <class F2>
BLS_FN void fp6_add(fp6_t<F2>& r, const fp6_t<F2>& a, const fp6_t<F2>& b) {
  fp2_add(r.a0, a.a0, b.a0);
  fp2_add(r.a1, a.a1, b.a1);
  fp2_add(r.a2, a.a2, b.a2);
}
template <class F2>
BLS_FN void fp6_sub(fp6_t<F2>& r, const fp6_t<F2>& a, const fp6_t<F2>& b) {
  fp2_sub(r.a0, a.a0, b.a0);
  fp2_sub(r.a1, a.a1, b.a1);
  fp2_sub(r.a2, a.a2, b.a2);
}
template <class F2>
BLS_FN void fp6_neg(fp6_t<F2>& r, const fp6_t<F2>& a) {
  fp2_neg(r.a0, a.a0);
  fp2_neg(r.a1, a.a1);
  fp2_neg(r.a2, a.a2);
}
template <class F2>
BLS_FN void fp6_norm(fp6_t<F2>& r, const fp6_t<F2>& a) {
  fp2_norm(r.a0, a.a0);
  fp2_norm(r.a1, a.a1);
  fp2_norm(r.a2, a.a2);
}
template <class F2>
BLS_FN void fp6_reduce(fp6_t<F2>& r, const fp6_t<F2>& a) {
  fp2_reduce(r.a0, a.a0);
  fp2_reduce(r.a1, a.a1);
  fp2_reduce(r.a2, a.a2);
}
// multiply by v: (a0, a1, a2) -> (xi a2, a0, a1)
template <class F2>
BLS_FN void fp6_mul_v(fp6_t<F2>& r, const fp6_t<F2>& a) {
  F2 t;
  fp2_mul_xi(t, a.a2);
  r.a2 = a.a1;
  r.a1 = a.a0;
  r.a0 = t;
}
// Karatsuba: 6 Fp2 multiplications.  Operands: normalised limbs (N+); result: normalised limbs, value up to 8 p.
template <class F2>
BLS_FN void fp6_mul(fp6_t<F2>& r, const fp6_t<F2>& a, const fp6_t<F2>& b) {
  F2 v0, v1, v2, s, t, m;
  fp2_mul(v0, a.a0, b.a0);
  fp2_mul(v1, a.a1, b.a1);
  fp2_mul(v2, a.a2, b.a2);
  fp6_t<F2> o;
  // c0 = v0 + xi((a1+a2)(b1+b2) - v1 - v2)
  fp2_add(s, a.a1, a.a2);
  fp2_add(t, b.a1, b.a2);
  fp2_mul(m, s, t);
  fp2_sub(m, m, v1);
  fp2_sub(m, m, v2);
  fp2_mul_xi(m, m);
  fp2_add(o.a0, m, v0);
  // c1 = (a0+a1)(b0+b1) - v0 - v1 + xi v2
  fp2_add(s, a.a0, a.a1);
  fp2_add(t, b.a0, b.a1);
  fp2_mul(m, s, t);
  fp2_sub(m, m, v0);
  fp2_sub(m, m, v1);
  fp2_mul_xi(t, v2);
  fp2_add(o.a1, m, t);
  // c2 = (a0+a2)(b0+b2) - v0 - v2 + v1
  fp2_add(s, a.a0, a.a2);
  fp2_add(t, b.a0, b.a2);
  fp2_mul(m, s, t);
  fp2_sub(m, m, v0);
  fp2_sub(m, m, v2);
  fp2_add(o.a2, m, v1);
  fp6_norm(r, o);
}
template <class F2>
BLS_FN void fp6_inv(fp6_t<F2>& r, const fp6_t<F2>& a) {
  F2 t0, t1, t2, m, d;
  fp2_sqr(t0, a.a0);
  fp2_mul(m, a.a1, a.a2);
  fp2_mul_xi(m, m);
  fp2_sub(t0, t0, m);  // a0^2 - xi a1 a2
  fp2_sqr(t1, a.a2);
  fp2_mul_xi(t1, t1);
  fp2_mul(m, a.a0, a.a1);
  fp2_sub(t1, t1, m);  // xi a2^2 - a0 a1
  fp2_sqr(t2, a.a1);
  fp2_mul(m, a.a0, a.a2);
  fp2_sub(t2, t2, m);  // a1^2 - a0 a2
  fp2_norm(t0, t0);
  fp2_norm(t1, t1);
  fp2_norm(t2, t2);
  fp2_mul(d, a.a0, t0);
  F2 e;
  fp2_mul(m, a.a2, t1);
  fp2_mul(e, a.a1, t2);
  fp2_add(m, m, e);
  fp2_mul_xi(m, m);
  fp2_add(d, d, m);
  fp2_reduce(d, d);
  fp2_inv(d, d);
  fp2_mul(r.a0, t0, d);
  fp2_mul(r.a1, t1, d);
  fp2_mul(r.a2, t2, d);
}

// ------------------------------------------------------------------ Fp12
template <class F2>
struct fp12_t {
  fp6_t<F2> c0, c1;
};
typedef fp12_t<fp2> fp12;

template <class F2>
BLS_FN void fp12_one(fp12_t<F2>& r) {
  fp6_zero(r.c0);
  fp6_zero(r.c1);
  fp2_one(r.c0.a0);
}
template <class F2>
BLS_FN bool fp12_is_one(const fp12_t<F2>& a) {
  F2 one;
  fp2_one(one);
  return fp2_eq(a.c0.a0, one) && fp2_is_zero(a.c0.a1) && fp2_is_zero(a.c0.a2) && fp2_is_zero(a.c1.a0) &&
         fp2_is_zero(a.c1.a1) && fp2_is_zero(a.c1.a2);
}
template <class F2>
BLS_FN void fp12_conj(fp12_t<F2>& r, const fp12_t<F2>& a) {
  r.c0 = a.c0;
  fp6_neg(r.c1, a.c1);
}
template <class F2>
BLS_FN void fp12_reduce(fp12_t<F2>& r, const fp12_t<F2>& a) {
  fp6_reduce(r.c0, a.c0);
  fp6_reduce(r.c1, a.c1);
}
// The four hot functions come as an inlinable NAME_body plus the non-inlined NAME the template code calls; the lane-split
// kernels also wrap the bodies in variants whose accumulator lives in LDS (tower_split.cuh).
// Fp12 functions take reduced or normalised operands (limbs N+, value within a few p) and return reduced results
// (exact limbs, value in (-0.52 p, 0.52 p)).
// Karatsuba over Fp6: 18 Fp2 multiplications
template <class F2>
BLS_FN void fp12_mul_body(fp12_t<F2>& r, const fp12_t<F2>& a, const fp12_t<F2>& b) {
  fp6_t<F2> t0, t1, s, t, m;
  fp6_mul(t0, a.c0, b.c0);
  fp6_mul(t1, a.c1, b.c1);
  fp6_add(s, a.c0, a.c1);
  fp6_add(t, b.c0, b.c1);
  fp6_norm(s, s);
  fp6_norm(t, t);
  fp6_mul(m, s, t);
  fp6_sub(m, m, t0);
  fp6_sub(m, m, t1);
  fp6_reduce(r.c1, m);
  fp6_mul_v(t1, t1);
  fp6_add(t0, t0, t1);
  fp6_reduce(r.c0, t0);
}
template <class F2>
BLS_NOINLINE void fp12_mul(fp12_t<F2>& r, const fp12_t<F2>& a, const fp12_t<F2>& b) {
  fp12_mul_body(r, a, b);
}
// complex squaring: 12 Fp2 multiplications
template <class F2>
BLS_FN void fp12_sqr_body(fp12_t<F2>& r, const fp12_t<F2>& a) {
  fp6_t<F2> t, s0, s1, m;
  fp6_mul(t, a.c0, a.c1);
  fp6_add(s0, a.c0, a.c1);
  fp6_mul_v(s1, a.c1);
  fp6_add(s1, s1, a.c0);
  fp6_norm(s0, s0);
  fp6_norm(s1, s1);
  fp6_mul(m, s0, s1);
  fp6_sub(m, m, t);
  fp6_mul_v(s0, t);
  fp6_sub(m, m, s0);
  fp6_reduce(r.c0, m);
  fp6_add(t, t, t);
  fp6_reduce(r.c1, t);
}
template <class F2>
BLS_NOINLINE void fp12_sqr(fp12_t<F2>& r, const fp12_t<F2>& a) {
  fp12_sqr_body(r, a);
}
template <class F2>
BLS_FN void fp12_inv_body(fp12_t<F2>& r, const fp12_t<F2>& a) {
  fp6_t<F2> t0, t1;
  fp6_mul(t0, a.c0, a.c0);
  fp6_mul(t1, a.c1, a.c1);
  fp6_mul_v(t1, t1);
  fp6_sub(t0, t0, t1);
  fp6_reduce(t0, t0);
  fp6_inv(t0, t0);
  fp6_reduce(t0, t0);
  fp6_mul(t1, a.c0, t0);
  fp6_reduce(r.c0, t1);
  fp6_mul(t1, a.c1, t0);
  fp6_neg(t1, t1);
  fp6_reduce(r.c1, t1);
}
template <class F2>
BLS_NOINLINE void fp12_inv(fp12_t<F2>& r, const fp12_t<F2>& a) {
  fp12_inv_body(r, a);
}
// a^(p^J), J = 1 or 2:  coefficient of w^k -> conj^J(c_k) * FROBJ[k]
template <int J, class F2>
BLS_FN void fp12_frob(fp12_t<F2>& r, const fp12_t<F2>& a) {
  const uint32_t(*tab)[2 * FP_NL] = (J == 1) ? FROB1 : FROB2;
  F2 c;
#define FROB_ONE(dst, src, k)        \
  if (J == 1) fp2_conj(c, src);      \
  else c = src;                      \
  if (k == 0) dst = c;               \
  else {                             \
    fp2_mul_const(dst, c, tab[k]);   \
  }
  FROB_ONE(r.c0.a0, a.c0.a0, 0)
  FROB_ONE(r.c1.a0, a.c1.a0, 1)
  FROB_ONE(r.c0.a1, a.c0.a1, 2)
  FROB_ONE(r.c1.a1, a.c1.a1, 3)
  FROB_ONE(r.c0.a2, a.c0.a2, 4)
  FROB_ONE(r.c1.a2, a.c1.a2, 5)
#undef FROB_ONE
}

// Granger-Scott squaring, valid in the cyclotomic subgroup (after the easy part of the final exponentiation)
template <class F2>
BLS_FN void fp4_sqr(F2& c0, F2& c1, const F2& a, const F2& b) {
  F2 t0, t1, t2;
  fp2_sqr(t0, a);
  fp2_sqr(t1, b);
  fp2_mul_xi(t2, t1);
  fp2_add(t2, t2, t0);
  fp2_norm(c0, t2);
  fp2_add(t2, a, b);
  fp2_norm(t2, t2);
  fp2_sqr(t2, t2);
  fp2_sub(t2, t2, t0);
  fp2_sub(t2, t2, t1);
  fp2_norm(c1, t2);
}
// ... with the two outputs left as the sums they are (limbs up to three products' worth): for callers that feed them to
// fp2_reduce_lin2 (the compressed squarings: 3 c +- 2 z in one pass)
template <class F2>
BLS_FN void fp4_sqr_lazy(F2& c0, F2& c1, const F2& a, const F2& b) {
  F2 t0, t1, t2;
  fp2_sqr(t0, a);
  fp2_sqr(t1, b);
  fp2_mul_xi(t2, t1);
  fp2_add(c0, t2, t0);
  fp2_add(t2, a, b);
  fp2_norm(t2, t2);
  fp2_sqr(t2, t2);
  fp2_sub(t2, t2, t0);
  fp2_sub(c1, t2, t1);
}
template <class F2>
BLS_FN void fp12_cyclotomic_sqr_body(fp12_t<F2>& r, const fp12_t<F2>& f) {
  F2 z0 = f.c0.a0, z4 = f.c0.a1, z3 = f.c0.a2, z2 = f.c1.a0, z1 = f.c1.a1, z5 = f.c1.a2;
  F2 t0, t1, t2, t3;
  fp4_sqr(t0, t1, z0, z1);
  fp2_sub(z0, t0, z0);
  fp2_dbl(z0, z0);
  fp2_add(z0, z0, t0);
  fp2_add(z1, t1, z1);
  fp2_dbl(z1, z1);
  fp2_add(z1, z1, t1);
  fp4_sqr(t0, t1, z2, z3);
  fp4_sqr(t2, t3, z4, z5);
  fp2_sub(z4, t0, z4);
  fp2_dbl(z4, z4);
  fp2_add(z4, z4, t0);
  fp2_add(z5, t1, z5);
  fp2_dbl(z5, z5);
  fp2_add(z5, z5, t1);
  fp2_mul_xi(t0, t3);
  fp2_norm(t0, t0);
  fp2_add(z2, t0, z2);
  fp2_dbl(z2, z2);
  fp2_add(z2, z2, t0);
  fp2_sub(z3, t2, z3);
  fp2_dbl(z3, z3);
  fp2_add(z3, z3, t2);
  fp2_reduce(r.c0.a0, z0);
  fp2_reduce(r.c0.a1, z4);
  fp2_reduce(r.c0.a2, z3);
  fp2_reduce(r.c1.a0, z2);
  fp2_reduce(r.c1.a1, z1);
  fp2_reduce(r.c1.a2, z5);
}
template <class F2>
BLS_NOINLINE void fp12_cyclotomic_sqr(fp12_t<F2>& r, const fp12_t<F2>& f) {
  fp12_cyclotomic_sqr_body(r, f);
}

// f * (l0 + l2 w^2 + l3 w^3): the sparse line value of the Miller loop.  13 Fp2 multiplications.
template <class F2>
BLS_FN void fp12_mul_by_line_body(fp12_t<F2>& f, const F2& l0, const F2& l2, const F2& l3) {
  // L0 = (l0, l2, 0), L1 = (0, l3, 0) in Fp6
  fp6_t<F2> t0, t1, s, m;
  F2 x, y, z;
  // t0 = f.c0 * (l0 + l2 v): Karatsuba on the two non-zero coefficients, 5 mul
  {
    const fp6_t<F2>& a = f.c0;
    F2 v0, v1;
    fp2_mul(v0, a.a0, l0);
    fp2_mul(v1, a.a1, l2);
    fp2_mul(x, a.a2, l2);  // a2 l2 v^3 = xi a2 l2
    fp2_mul_xi(x, x);
    fp2_add(t0.a0, v0, x);
    fp2_add(y, a.a0, a.a1);
    fp2_add(z, l0, l2);
    fp2_mul(y, y, z);
    fp2_sub(y, y, v0);
    fp2_sub(t0.a1, y, v1);  // a0 l2 + a1 l0
    fp2_mul(z, a.a2, l0);
    fp2_add(t0.a2, z, v1);  // a2 l0 + a1 l2
    fp6_norm(t0, t0);
  }
  // t1 = f.c1 * (l3 v) = (xi a2 l3, a0 l3, a1 l3), 3 mul
  {
    const fp6_t<F2>& a = f.c1;
    fp2_mul(x, a.a2, l3);
    fp2_mul_xi(t1.a0, x);
    fp2_mul(t1.a1, a.a0, l3);
    fp2_mul(t1.a2, a.a1, l3);
  }
  // m = (f.c0 + f.c1) * (l0 + (l2 + l3) v), 5 mul
  fp6_add(s, f.c0, f.c1);
  fp6_norm(s, s);
  {
    F2 l23, v0, v1;
    fp2_add(l23, l2, l3);
    fp2_norm(l23, l23);
    fp2_mul(v0, s.a0, l0);
    fp2_mul(v1, s.a1, l23);
    fp2_mul(x, s.a2, l23);
    fp2_mul_xi(x, x);
    fp2_add(m.a0, v0, x);
    fp2_add(y, s.a0, s.a1);
    fp2_add(z, l0, l23);
    fp2_mul(y, y, z);
    fp2_sub(y, y, v0);
    fp2_sub(m.a1, y, v1);
    fp2_mul(z, s.a2, l0);
    fp2_add(m.a2, z, v1);
  }
  fp6_sub(m, m, t0);
  fp6_sub(m, m, t1);
  fp6_reduce(f.c1, m);
  fp6_mul_v(t1, t1);
  fp6_add(t0, t0, t1);
  fp6_reduce(f.c0, t0);
}
template <class F2>
BLS_NOINLINE void fp12_mul_by_line(fp12_t<F2>& f, const F2& l0, const F2& l2, const F2& l3) {
  fp12_mul_by_line_body(f, l0, l2, l3);
}

// x * (y1 v + y2 v^2) in Fp6: 5 Fp2 multiplications.  Operands and result as fp6_mul.
template <class F2>
BLS_FN void fp6_mul_by_12(fp6_t<F2>& r, const fp6_t<F2>& x, const F2& y1, const F2& y2) {
  F2 v1, v2, s, t, m, o0, o1, o2;
  fp2_mul(v1, x.a1, y1);
  fp2_mul(v2, x.a2, y2);
  fp2_add(s, x.a1, x.a2);
  fp2_add(t, y1, y2);
  fp2_mul(m, s, t);
  fp2_sub(m, m, v1);
  fp2_sub(m, m, v2);        // x1 y2 + x2 y1
  fp2_mul_xi(o0, m);
  fp2_mul(m, x.a0, y1);
  fp2_mul_xi(t, v2);
  fp2_add(o1, m, t);        // x0 y1 + xi x2 y2
  fp2_mul(m, x.a0, y2);
  fp2_add(o2, m, v1);       // x0 y2 + x1 y1
  fp2_norm(r.a0, o0);
  fp2_norm(r.a1, o1);
  fp2_norm(r.a2, o2);
}
// ---- merged line values.  (a0 + a2 w^2 + a3 w^3)(b0 + b2 w^2 + b3 w^3) = c0 + c2 w^2 + c3 w^3 + c4 w^4 + c5 w^5: the two line
// values of one Miller step multiplied together first (6 Fp2 multiplications; the product has no w^1 term), then ONE Fp12
// multiplication with that structure (6 + 5 + 6): 23 Fp2 multiplications instead of 2 x 13, and the accumulator is read and
// written once.  The coefficients leave REDUCED (c4 is a plain product): the Karatsuba sums of the Fp12 product double values
// twice over.  Round 3 splits the two halves between two kernels (kernels.cuh k_lines2s / k_millerf2s): the kernel that walks
// the G2 point merges the lines and streams the five coefficients through HBM; the accumulator's kernel holds nothing but f.
template <class F2>
struct line5_t {
  F2 c0, c2, c4, c3, c5;
};
template <class F2>
BLS_FN void lines_merge(line5_t<F2>& L, const F2& a0, const F2& a2, const F2& a3, const F2& b0, const F2& b2, const F2& b3) {
  F2 p00, p22, p33, s, t, m;
  fp2_mul(p00, a0, b0);
  fp2_mul(p22, a2, b2);
  fp2_mul(p33, a3, b3);
  fp2_mul_xi(m, p33);
  fp2_add(m, m, p00);
  fp2_reduce(L.c0, m);       // c0 = a0 b0 + xi a3 b3
  fp2_add(s, a0, a2);
  fp2_add(t, b0, b2);
  fp2_mul(m, s, t);
  fp2_sub(m, m, p00);
  fp2_sub(m, m, p22);
  fp2_reduce(L.c2, m);       // c2 = a0 b2 + a2 b0
  L.c4 = p22;              // c4
  fp2_add(s, a0, a3);
  fp2_add(t, b0, b3);
  fp2_mul(m, s, t);
  fp2_sub(m, m, p00);
  fp2_sub(m, m, p33);
  fp2_reduce(L.c3, m);       // c3 = a0 b3 + a3 b0
  fp2_add(s, a2, a3);
  fp2_add(t, b2, b3);
  fp2_mul(m, s, t);
  fp2_sub(m, m, p22);
  fp2_sub(m, m, p33);
  fp2_reduce(L.c5, m);       // c5 = a2 b3 + a3 b2
}
// the same with b3 = y in Fp (a line of a FIXED G2 argument, scaled so that its w^3 coefficient is yP itself: g2neg_lines.cuh
// *_LINES_N): a3 b3 is an Fp2-by-Fp product (one product stream per lane instead of two)
template <class F2>
BLS_FN void lines_merge_y(line5_t<F2>& L, const F2& a0, const F2& a2, const F2& a3, const F2& b0, const F2& b2, const fp& y) {
  F2 p00, p22, p33, s, t, m, yy;
  fp2_from_fp(yy, y);
  fp2_mul(p00, a0, b0);
  fp2_mul(p22, a2, b2);
  fp2_mul_fp(p33, a3, y);
  fp2_mul_xi(m, p33);
  fp2_add(m, m, p00);
  fp2_reduce(L.c0, m);
  fp2_add(s, a0, a2);
  fp2_add(t, b0, b2);
  fp2_mul(m, s, t);
  fp2_sub(m, m, p00);
  fp2_sub(m, m, p22);
  fp2_reduce(L.c2, m);
  L.c4 = p22;
  fp2_add(s, a0, a3);
  fp2_add(t, b0, yy);
  fp2_mul(m, s, t);
  fp2_sub(m, m, p00);
  fp2_sub(m, m, p33);
  fp2_reduce(L.c3, m);
  fp2_add(s, a2, a3);
  fp2_add(t, b2, yy);
  fp2_mul(m, s, t);
  fp2_sub(m, m, p22);
  fp2_sub(m, m, p33);
  fp2_reduce(L.c5, m);
}
// f * (c0 + c2 w^2 + c3 w^3 + c4 w^4 + c5 w^5): L0 = (c0, c2, c4), L1 = (0, c3, c5) in Fp6; 6 + 5 + 6 Fp2 multiplications
template <class F2>
BLS_FN void fp12_mul_by_line5_body(fp12_t<F2>& f, const line5_t<F2>& L) {
  fp6_t<F2> t0, t1, s, m, L0, Ls;
  L0.a0 = L.c0;
  L0.a1 = L.c2;
  L0.a2 = L.c4;
  fp6_mul(t0, f.c0, L0);
  fp6_mul_by_12(t1, f.c1, L.c3, L.c5);
  fp6_add(s, f.c0, f.c1);
  fp6_norm(s, s);
  Ls.a0 = L.c0;
  fp2_add(Ls.a1, L.c2, L.c3);
  fp2_add(Ls.a2, L.c4, L.c5);
  fp2_norm(Ls.a1, Ls.a1);
  fp2_norm(Ls.a2, Ls.a2);
  fp6_mul(m, s, Ls);
  fp6_sub(m, m, t0);
  fp6_sub(m, m, t1);
  fp6_reduce(f.c1, m);
  fp6_mul_v(t1, t1);
  fp6_add(t0, t0, t1);
  fp6_reduce(f.c0, t0);
}
template <class F2>
BLS_FN void fp12_mul_by_2lines_body(fp12_t<F2>& f, const F2& a0, const F2& a2, const F2& a3, const F2& b0, const F2& b2, const F2& b3) {
  line5_t<F2> L;
  lines_merge(L, a0, a2, a3, b0, b2, b3);
  fp12_mul_by_line5_body(f, L);
}
// the Fp12 element of a merged line value (the accumulator's first value: 1 * L)
template <class F2>
BLS_FN void fp12_from_line5(fp12_t<F2>& f, const line5_t<F2>& L) {
  f.c0.a0 = L.c0;
  f.c0.a1 = L.c2;
  f.c0.a2 = L.c4;
  fp2_zero(f.c1.a0);
  f.c1.a1 = L.c3;
  f.c1.a2 = L.c5;
}
template <class F2>
BLS_NOINLINE void fp12_mul_by_2lines(fp12_t<F2>& f, const F2& a0, const F2& a2, const F2& a3, const F2& b0, const F2& b2, const F2& b3) {
  fp12_mul_by_2lines_body(f, a0, a2, a3, b0, b2, b3);
}
