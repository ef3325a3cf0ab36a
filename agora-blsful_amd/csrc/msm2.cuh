// Multi-scalar multiplication, second generation: sum_i k_i P_i -- the loop `aggregated_pk += pk.0 * *coeff` of reference
// src/secure_aggregation.rs:201-204 (and its sign-side twin :163-166) as a bucket method built for this machine.
//
//   * Endomorphism split.  On G2 psi(P) = [x]P (x = -z, z = 0xd201000000010000), so with the base-z digits of the scalar
//       k = a0 + a1 z + a2 z^2 + a3 z^3     (a_j < z < 2^64)
//     k P = a0 P + a1 (-psi P) + a2 (psi^2 P) + a3 (-psi^3 P): four 64-bit scalars on four cheap images of the point.
//     On G1 phi(P) = (beta x, y) = [-z^2]P, so k = a0 + a1 z^2 gives k P = a0 P + a1 (-phi P): two 128-bit scalars.
//     The chains of doublings that weigh the windows -- the critical path of the whole sum on a GPU, where ONE doubling
//     costs microseconds of latency whatever the occupancy -- shrink from 255 to 64 (G2) / 128 (G1) doublings.
//   * Signed window digits in (-2^(c-1), 2^(c-1)]: half the buckets; a negative digit adds the negated point.
//   * Mixed additions: every point is brought to affine once (its images with it), so a bucket addition is the
//     8M + 3S mixed form instead of 11M + 5S.
//   * The part that does not depend on the scalars (affine conversion + images, k_msm2_prep) is a separate launch:
//     verify_secure runs it while a host core still hashes the sorted key stream.
// The result is the same group element as the reference's serial loop; it leaves with Z = 1, so the bytes do not depend on
// the bucket fill order.  (tests: tests/test_hostsim.py for the per-item functions with the bound tracker on;
// tests/test_gpu_api.py::test_msm_pippenger_closed_form, test_sum_and_msm and the config-5 full-size tests on the GPU.)
#pragma once

// ---- mixed addition r = p + (x2, y2), the second operand affine and not the identity (madd-2007-bl, exceptional cases
// handled: attacker-chosen inputs do produce equal and opposite points)
template <class F>
BLS_FN void jac_madd_body(jac<F>& r, const jac<F>& p, const F& x2, const F& y2) {
  if (jac_is_inf(p)) {
    r.x = x2;
    r.y = y2;
    fe_one(r.z);
    return;
  }
  F z1z1, u2, s2, h, i, j, rr, v, t, x3, y3, z3;
  fe_sqr(z1z1, p.z);
  fe_mul(u2, x2, z1z1);
  fe_mul(s2, y2, p.z);
  fe_mul(s2, s2, z1z1);
  fe_sub(h, u2, p.x);
  fe_sub(rr, s2, p.y);
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr)) {
      jac<F> q;
      q.x = x2;
      q.y = y2;
      fe_one(q.z);
      jac_dbl(r, q);
    } else {
      jac_set_inf(r);
    }
    return;
  }
  fe_dbl(rr, rr);
  fe_reduce(rr, rr);
  fe_dbl(i, h);
  fe_reduce(i, i);
  fe_sqr(i, i);
  fe_mul(j, h, i);
  fe_mul(v, p.x, i);
  fe_sqr(x3, rr);
  fe_sub(x3, x3, j);
  fe_dbl(t, v);
  fe_sub(x3, x3, t);
  fe_reduce(x3, x3);
  fe_sub(t, v, x3);
  fe_mul(y3, rr, t);
  fe_mul(t, p.y, j);
  fe_dbl(t, t);
  fe_sub(y3, y3, t);
  fe_reduce(h, h);
  fe_mul(z3, p.z, h);
  fe_dbl(z3, z3);          // Z3 = 2 Z1 H
  r.x = x3;
  fe_reduce(r.y, y3);
  fe_reduce(r.z, z3);
}
template <class F>
BLS_NOINLINE void jac_madd(jac<F>& r, const jac<F>& p, const F& x2, const F& y2) {
  jac_madd_body(r, p, x2, y2);
}

// ---- scalar decomposition.  k: 8 little-endian 32-bit words, any 256-bit value (reduced modulo r first).
BLS_FN void msm2_mod_r(uint32_t v[8]) {
  const uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  for (int round = 0; round < 3; round++) {   // 2^256 < 2.3 r
    uint32_t d[8];
    uint64_t bw = 0;
    for (int j = 0; j < 8; j++) {
      const uint64_t t = (uint64_t)v[j] - R[j] - bw;
      d[j] = (uint32_t)t;
      bw = (t >> 63) & 1;
    }
    if (!bw)
      for (int j = 0; j < 8; j++) v[j] = d[j];
  }
}
// q = v / z, returns v mod z (z = |x| of the curve, 64 bits); bit-serial restoring division: 256 steps of shift / compare /
// subtract -- a few thousand instructions per scalar against the hundreds of thousands of its bucket additions
BLS_FN uint64_t msm2_divrem_z(uint32_t q[8], const uint32_t v[8]) {
  const uint64_t z = BLS_X_ABS;
  uint64_t rem = 0;
  for (int j = 0; j < 8; j++) q[j] = 0;
  for (int bit = 255; bit >= 0; bit--) {
    const uint64_t top = rem >> 63;
    rem = (rem << 1) | ((v[bit >> 5] >> (bit & 31)) & 1u);
    if (top || rem >= z) {
      rem -= z;               // with top set the true value is rem + 2^64 >= z: the wrapped subtraction is exact
      q[bit >> 5] |= 1u << (bit & 31);
    }
  }
  return rem;
}
// G2: a[0..3] with k = a0 + a1 z + a2 z^2 + a3 z^3
BLS_FN void msm2_decompose_g2(uint64_t a[4], const uint32_t* k) {
  uint32_t v[8], q[8];
  for (int j = 0; j < 8; j++) v[j] = k[j];
  msm2_mod_r(v);
  a[0] = msm2_divrem_z(q, v);
  a[1] = msm2_divrem_z(v, q);
  a[2] = msm2_divrem_z(q, v);
  a[3] = (uint64_t)q[0] | ((uint64_t)q[1] << 32);   // < r / z^3 < 2^64
}
// G1: a[0..1] = a0 (low, high 64 bits), a[2..3] = a1 with k = a0 + a1 z^2: two divisions by z, a0 = r0 + r1 z
BLS_FN void msm2_decompose_g1(uint64_t a[4], const uint32_t* k) {
  uint32_t v[8], q[8];
  for (int j = 0; j < 8; j++) v[j] = k[j];
  msm2_mod_r(v);
  const uint64_t r0 = msm2_divrem_z(q, v);
  const uint64_t r1 = msm2_divrem_z(v, q);        // v = k / z^2 < 2^128
  // a0 = r0 + r1 z  (< z^2 < 2^128): 64 x 64 -> 128-bit product by 32-bit halves
  const uint64_t z = BLS_X_ABS;
  const uint64_t zl = z & 0xffffffffu, zh = z >> 32, rl = r1 & 0xffffffffu, rh = r1 >> 32;
  const uint64_t p0 = rl * zl, p1 = rl * zh, p2 = rh * zl, p3 = rh * zh;
  const uint64_t mid = (p0 >> 32) + (p1 & 0xffffffffu) + (p2 & 0xffffffffu);
  uint64_t lo = (p0 & 0xffffffffu) | (mid << 32);
  uint64_t hi = p3 + (p1 >> 32) + (p2 >> 32) + (mid >> 32);
  const uint64_t lo2 = lo + r0;
  hi += lo2 < lo ? 1 : 0;
  a[0] = lo2;
  a[1] = hi;
  a[2] = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
  a[3] = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
}
// Window layout: the bits + 1 bit positions of a sub-scalar (one spare for the carry of the signed recoding) are spread over W
// windows whose widths differ by at most one bit (the first `rem` windows are one bit wider), so that no window is a
// narrow remainder whose few buckets would each receive a large share of all points.
struct msm2_layout {
  int total, W, base, rem;     // total = bits + 1; width(w) = base + (w < rem)
};
BLS_FN msm2_layout msm2_make_layout(int bits, int W) {
  msm2_layout L;
  L.total = bits + 1;
  L.W = W;
  L.base = L.total / W;
  L.rem = L.total % W;
  return L;
}
BLS_FN int msm2_width(const msm2_layout& L, int w) { return L.base + (w < L.rem ? 1 : 0); }
BLS_FN int msm2_start(const msm2_layout& L, int w) { return w * L.base + (w < L.rem ? w : L.rem); }
// first bucket of window w (a window of width c has 2^(c-1) buckets: digit magnitudes 1 .. 2^(c-1))
BLS_FN size_t msm2_bucket_base(const msm2_layout& L, int w) {
  const int wide = w < L.rem ? w : L.rem;
  return ((size_t)wide << L.base) + ((size_t)(w - wide) << (L.base - 1));
}
BLS_FN size_t msm2_buckets(const msm2_layout& L) { return msm2_bucket_base(L, L.W); }
// the window of bucket b (inverse of msm2_bucket_base: the first `rem` windows have 2^base buckets each, the others 2^(base-1))
BLS_FN int msm2_window_of(const msm2_layout& L, size_t b) {
  const size_t wide = (size_t)L.rem << L.base;
  if (b < wide) return (int)(b >> L.base);
  return L.rem + (int)((b - wide) >> (L.base - 1));
}
// signed digit of window w of a sub-scalar given as `words` 64-bit words: digits in (-2^(c-1), 2^(c-1)], recoded from the
// low end (digit w needs the carry of digit w - 1, so the caller walks w upwards and threads `carry` through)
BLS_FN int32_t msm2_digit(const uint64_t* a, int words, const msm2_layout& L, int w, uint32_t& carry) {
  const int bit = msm2_start(L, w), c = msm2_width(L, w), wi = bit >> 6, sh = bit & 63;
  uint64_t v = wi < words ? a[wi] >> sh : 0;
  if (sh && wi + 1 < words) v |= a[wi + 1] << (64 - sh);
  int32_t d = (int32_t)((uint32_t)v & ((1u << c) - 1u)) + (int32_t)carry;
  if (d > (1 << (c - 1))) {
    d -= 1 << c;
    carry = 1;
  } else {
    carry = 0;
  }
  return d;
}

// ---- the scalar-independent images of an affine point, with the signs of the decomposition folded in
//   G1: Q0 = P, Q1 = -phi(P) = (beta x, -y)
//   G2: Q0 = P, Q1 = -psi(P), Q2 = psi^2(P), Q3 = -psi^3(P)
BLS_FN void msm2_images_g1(fp qx[2], fp qy[2], const g1_aff& p) {
  fp beta;
  fp_load(beta, G1_BETA);
  fp_reduce(qx[0], p.x);
  fp_reduce(qy[0], p.y);
  fp_mul(qx[1], p.x, beta);
  fp_neg(qy[1], p.y);
  fp_reduce(qy[1], qy[1]);
}
BLS_FN void msm2_images_g2(fp2 qx[4], fp2 qy[4], const g2_aff& p) {
  fp cx2, cy2;
  fp2 t;
  fp_load(cx2, PSI2_CX);
  fp_load(cy2, PSI2_CY);
  fp2_reduce(qx[0], p.x);
  fp2_reduce(qy[0], p.y);
  fp2_conj(t, p.x);
  fp2_mul_const(qx[1], t, PSI_CX);
  fp2_conj(t, p.y);
  fp2_mul_const(t, t, PSI_CY);
  fp2_neg(t, t);
  fp2_reduce(qy[1], t);                 // -psi(P)
  fp2_mul_fp(qx[2], p.x, cx2);
  fp2_mul_fp(qy[2], p.y, cy2);          // psi^2(P)
  fp2_conj(t, qx[2]);
  fp2_mul_const(qx[3], t, PSI_CX);
  fp2_conj(t, qy[2]);
  fp2_mul_const(t, t, PSI_CY);
  fp2_neg(t, t);
  fp2_reduce(qy[3], t);                 // -psi(psi^2 P)
}
