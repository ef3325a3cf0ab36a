// G1 (over Fp) and G2 (over Fp2, the M-twist y^2 = x^3 + 4(1+u)) group arithmetic in Jacobian coordinates
// (x = X/Z^2, y = Y/Z^3, infinity = Z == 0) -- the coordinate system of blst's blst_p1/blst_p2, which is what the
// reference's G1Projective/G2Projective hold in memory: RAW_PROJ buffers need only the per-coordinate
// radix change of fp_from_raw / fp_to_raw.
// Bounds (fp.cuh): point coordinates enter and leave every function here reduced (exact limbs, |value| < 1.2 p).
// Replaces: `+`, `-`, `* scalar`, `to_affine`, `is_identity`, `to_compressed` of the un-vendored backend
// (reference call sites src/traits/pk_multi.rs:10, src/secure_aggregation.rs:42,203, src/helpers.rs:44,56).
#pragma once
#include "tower.cuh"

// ---- uniform field interface so point code is written once
BLS_FN void fe_add(fp& r, const fp& a, const fp& b) { fp_add(r, a, b); }
BLS_FN void fe_sub(fp& r, const fp& a, const fp& b) { fp_sub(r, a, b); }
BLS_FN void fe_mul(fp& r, const fp& a, const fp& b) { fp_mul(r, a, b); }
BLS_FN void fe_sqr(fp& r, const fp& a) { fp_sqr(r, a); }
BLS_FN void fe_neg(fp& r, const fp& a) { fp_neg(r, a); }
BLS_FN void fe_dbl(fp& r, const fp& a) { fp_dbl(r, a); }
BLS_FN void fe_inv(fp& r, const fp& a) { fp_inv(r, a); }
BLS_FN bool fe_is_zero(const fp& a) { return fp_is_zero(a); }
BLS_FN bool fe_eq(const fp& a, const fp& b) { return fp_eq(a, b); }
BLS_FN void fe_zero(fp& r) { fp_zero(r); }
BLS_FN void fe_one(fp& r) { fp_one(r); }
BLS_FN void fe_cmov(fp& r, const fp& a, bool c) { fp_cmov(r, a, c); }
BLS_FN void fe_norm(fp& r, const fp& a) { fp_norm(r, a); }
BLS_FN void fe_reduce(fp& r, const fp& a) { fp_reduce(r, a); }

BLS_FN void fe_add(fp2& r, const fp2& a, const fp2& b) { fp2_add(r, a, b); }
BLS_FN void fe_sub(fp2& r, const fp2& a, const fp2& b) { fp2_sub(r, a, b); }
BLS_FN void fe_mul(fp2& r, const fp2& a, const fp2& b) { fp2_mul(r, a, b); }
BLS_FN void fe_sqr(fp2& r, const fp2& a) { fp2_sqr(r, a); }
BLS_FN void fe_neg(fp2& r, const fp2& a) { fp2_neg(r, a); }
BLS_FN void fe_dbl(fp2& r, const fp2& a) { fp2_dbl(r, a); }
BLS_FN void fe_inv(fp2& r, const fp2& a) { fp2_inv(r, a); }
BLS_FN bool fe_is_zero(const fp2& a) { return fp2_is_zero(a); }
BLS_FN bool fe_eq(const fp2& a, const fp2& b) { return fp2_eq(a, b); }
BLS_FN void fe_zero(fp2& r) { fp2_zero(r); }
BLS_FN void fe_one(fp2& r) { fp2_one(r); }
BLS_FN void fe_cmov(fp2& r, const fp2& a, bool c) { fp2_cmov(r, a, c); }
BLS_FN void fe_norm(fp2& r, const fp2& a) { fp2_norm(r, a); }
BLS_FN void fe_reduce(fp2& r, const fp2& a) { fp2_reduce(r, a); }

template <class F>
struct jac {
  F x, y, z;
};
template <class F>
struct aff {
  F x, y;
  bool inf;
};
typedef jac<fp> g1_jac;
typedef jac<fp2> g2_jac;
typedef aff<fp> g1_aff;
typedef aff<fp2> g2_aff;

template <class F>
BLS_FN void jac_set_inf(jac<F>& r) {
  fe_one(r.x);
  fe_one(r.y);
  fe_zero(r.z);
}
template <class F>
BLS_FN bool jac_is_inf(const jac<F>& a) {
  return fe_is_zero(a.z);
}
template <class F>
BLS_FN void jac_neg(jac<F>& r, const jac<F>& a) {
  r.x = a.x;
  fe_neg(r.y, a.y);
  r.z = a.z;
}
template <class F>
BLS_FN void jac_from_aff(jac<F>& r, const aff<F>& a) {
  if (a.inf) {
    jac_set_inf(r);
    return;
  }
  r.x = a.x;
  r.y = a.y;
  fe_one(r.z);
}

// The point functions are not inlined: each is a few thousand instructions, and the prepare / MSM kernels call them
// from dozens of sites (inlining all of them made single translation units compile for the better part of an hour).
// doubling on y^2 = x^3 + b (a = 0), dbl-2009-l: 2M + 5S
template <class F>
BLS_FN void jac_dbl_body(jac<F>& r, const jac<F>& p) {
  F A, B, C, D, E, Fq, t;
  fe_sqr(A, p.x);
  fe_sqr(B, p.y);
  fe_sqr(C, B);
  fe_add(t, p.x, B);
  fe_sqr(t, t);
  fe_sub(t, t, A);
  fe_sub(t, t, C);
  fe_dbl(D, t);
  fe_reduce(D, D);       // D = 2((X + B)^2 - A - C)
  fe_dbl(E, A);
  fe_add(E, E, A);
  fe_reduce(E, E);       // E = 3A
  fe_sqr(Fq, E);
  F z3;
  fe_mul(z3, p.y, p.z);
  fe_dbl(z3, z3);
  fe_dbl(t, D);
  fe_sub(t, Fq, t);
  F x3;
  fe_reduce(x3, t);      // X3 = F - 2D
  fe_sub(t, D, x3);
  fe_mul(t, E, t);
  fe_dbl(C, C);
  fe_dbl(C, C);
  fe_reduce(C, C);
  fe_dbl(C, C);          // 8C
  fe_sub(t, t, C);
  fe_reduce(r.y, t);
  r.x = x3;
  fe_reduce(r.z, z3);
}
template <class F>
BLS_NOINLINE void jac_dbl(jac<F>& r, const jac<F>& p) {
  jac_dbl_body(r, p);
}

// general addition, add-2007-bl with the exceptional cases handled (inputs are attacker-chosen: equal or opposite
// points do occur, e.g. duplicated public keys in an aggregate)
template <class F>
BLS_FN void jac_add_body(jac<F>& r, const jac<F>& p, const jac<F>& q) {
  if (jac_is_inf(p)) {
    r = q;
    return;
  }
  if (jac_is_inf(q)) {
    r = p;
    return;
  }
  F z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
  fe_sqr(z1z1, p.z);
  fe_sqr(z2z2, q.z);
  fe_mul(u1, p.x, z2z2);
  fe_mul(u2, q.x, z1z1);
  fe_mul(s1, p.y, q.z);
  fe_mul(s1, s1, z2z2);
  fe_mul(s2, q.y, p.z);
  fe_mul(s2, s2, z1z1);
  fe_sub(h, u2, u1);
  fe_sub(rr, s2, s1);
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr)) {
      jac_dbl(r, p);
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
  fe_mul(v, u1, i);
  F x3, y3, z3;
  fe_sqr(x3, rr);
  fe_sub(x3, x3, j);
  fe_dbl(t, v);
  fe_sub(x3, x3, t);
  fe_reduce(x3, x3);
  fe_sub(t, v, x3);
  fe_mul(y3, rr, t);
  fe_mul(t, s1, j);
  fe_dbl(t, t);
  fe_sub(y3, y3, t);
  fe_add(z3, p.z, q.z);
  fe_sqr(z3, z3);
  fe_sub(z3, z3, z1z1);
  fe_sub(z3, z3, z2z2);
  fe_reduce(z3, z3);
  fe_reduce(h, h);
  fe_mul(z3, z3, h);
  r.x = x3;
  fe_reduce(r.y, y3);
  r.z = z3;
}
template <class F>
BLS_NOINLINE void jac_add(jac<F>& r, const jac<F>& p, const jac<F>& q) {
  jac_add_body(r, p, q);
}

// mixed addition with an affine second operand (Z2 = 1)
template <class F>
BLS_FN void jac_add_aff(jac<F>& r, const jac<F>& p, const aff<F>& q) {
  jac<F> qq;
  jac_from_aff(qq, q);
  jac_add(r, p, qq);
}

template <class F>
BLS_NOINLINE void jac_to_aff(aff<F>& r, const jac<F>& p) {
  if (jac_is_inf(p)) {
    r.inf = true;
    fe_zero(r.x);
    fe_zero(r.y);
    return;
  }
  F zi, zi2;
  fe_inv(zi, p.z);
  fe_sqr(zi2, zi);
  fe_mul(r.x, p.x, zi2);
  fe_mul(zi2, zi2, zi);
  fe_mul(r.y, p.y, zi2);
  r.inf = false;
}

// [k] P for a 64-bit public scalar (left-to-right; k is the same in every lane: no divergence)
template <class F>
BLS_NOINLINE void jac_mul_u64(jac<F>& r, const jac<F>& p, uint64_t k) {
  jac<F> acc;
  jac_set_inf(acc);
  for (int i = 63; i >= 0; i--) {
    jac_dbl_body(acc, acc);                       // inlined here: the accumulator stays in registers across the 64 steps
    if ((k >> i) & 1) jac_add_body(acc, acc, p);
  }
  r = acc;
}

// [hi 2^64 + lo] P (the 128-bit scalars of the opt-in grouped verification: per-lane values, the branch diverges)
template <class F>
BLS_NOINLINE void jac_mul_u128(jac<F>& r, const jac<F>& p, uint64_t hi, uint64_t lo) {
  jac<F> acc;
  jac_set_inf(acc);
  for (int i = 127; i >= 0; i--) {
    jac_dbl_body(acc, acc);
    if (((i >= 64 ? hi : lo) >> (i & 63)) & 1) jac_add_body(acc, acc, p);
  }
  r = acc;
}

// [k] P for a 256-bit scalar given as 8 little-endian 32-bit words (per-lane scalars: the branch diverges)
template <class F>
BLS_NOINLINE void jac_mul_scalar(jac<F>& r, const jac<F>& p, const uint32_t* k) {
  jac<F> acc;
  jac_set_inf(acc);
  for (int i = 255; i >= 0; i--) {
    jac_dbl(acc, acc);
    if ((k[i >> 5] >> (i & 31)) & 1) jac_add(acc, acc, p);
  }
  r = acc;
}

// ---- psi (untwist-Frobenius-twist) on G2, Jacobian form: (conj X * cx, conj Y * cy, conj Z).  Templates over the Fp2
// representation: the one-lane fp2 and the lane-split hfp2 (tower_split.cuh), whose latency mode halves these steps.
template <class F2>
BLS_FN void g2_psi(jac<F2>& r, const jac<F2>& p) {
  F2 t;
  fp2_conj(t, p.x);
  fp2_mul_const(r.x, t, PSI_CX);
  fp2_conj(t, p.y);
  fp2_mul_const(r.y, t, PSI_CY);
  fp2_conj(r.z, p.z);
}
// psi^2: (X * cx2, Y * cy2, Z) with cx2, cy2 in Fp
template <class F2>
BLS_FN void g2_psi2(jac<F2>& r, const jac<F2>& p) {
  fp cx2, cy2;
  fp_load(cx2, PSI2_CX);
  fp_load(cy2, PSI2_CY);
  fp2_mul_fp(r.x, p.x, cx2);
  fp2_mul_fp(r.y, p.y, cy2);
  r.z = p.z;
}

// RFC 9380 Appendix G.3 clear_cofactor_bls12381_g2 (x is negative: [x]P = -[|x|]P)
template <class F2>
BLS_NOINLINE void g2_clear_cofactor(jac<F2>& r, const jac<F2>& p) {
  jac<F2> t1, t2, t3, n;
  jac_mul_u64(t1, p, BLS_X_ABS);
  jac_neg(t1, t1);  // t1 = x P
  g2_psi(t2, p);    // t2 = psi(P)
  jac_dbl(t3, p);
  g2_psi2(t3, t3);  // t3 = psi^2(2P)
  jac_neg(n, t2);
  jac_add(t3, t3, n);  // t3 - t2
  jac_add(t2, t1, t2);  // t1 + t2
  jac_mul_u64(t2, t2, BLS_X_ABS);
  jac_neg(t2, t2);      // x (t1 + t2)
  jac_add(t3, t3, t2);
  jac_neg(n, t1);
  jac_add(t3, t3, n);
  jac_neg(n, p);
  jac_add(r, t3, n);
}

// ---- ZCash compressed encoding (modern); `legacy` applies the Dash header transcode of
// reference src/impls/legacy.rs:19-35
BLS_FN void fp_to_be48(uint8_t* out, const fp& a_mont) {
  fp t;
  uint32_t ww[12];
  fp_from_mont(t, a_mont);
  fp_get_words(ww, t);
#pragma unroll
  for (int i = 0; i < 12; i++) {
    uint32_t w = ww[11 - i];
    out[4 * i + 0] = (uint8_t)(w >> 24);
    out[4 * i + 1] = (uint8_t)(w >> 16);
    out[4 * i + 2] = (uint8_t)(w >> 8);
    out[4 * i + 3] = (uint8_t)w;
  }
}
BLS_FN void header_to_legacy(uint8_t* b) {
  if (b[0] == 0xc0) return;
  uint8_t ys = b[0] & 0x20;
  b[0] &= 0x1f;
  if (ys) b[0] |= 0x80;
}
BLS_FN void g1_compress(uint8_t* out, const g1_aff& a, bool legacy) {
  if (a.inf) {
    for (int i = 0; i < 48; i++) out[i] = 0;
    out[0] = 0xc0;
    return;
  }
  fp_to_be48(out, a.x);
  out[0] |= 0x80 | (fp_lex_largest(a.y) ? 0x20 : 0);
  if (legacy) header_to_legacy(out);
}
BLS_FN void g2_compress(uint8_t* out, const g2_aff& a, bool legacy) {
  if (a.inf) {
    for (int i = 0; i < 96; i++) out[i] = 0;
    out[0] = 0xc0;
    return;
  }
  fp_to_be48(out, a.x.c1);
  fp_to_be48(out + 48, a.x.c0);
  out[0] |= 0x80 | (fp2_lex_largest(a.y) ? 0x20 : 0);
  if (legacy) header_to_legacy(out);
}

// ---- subgroup membership (endomorphism tests, Scott 2021): needed when points arrive as wire bytes
// G1: phi(P) = (beta x, y) == [-x^2] P          G2: psi(P) == [x] P          (x = -|x|)
BLS_NOINLINE bool g1_in_subgroup(const g1_aff& p) {
  if (p.inf) return true;
  g1_jac j, q;
  jac_from_aff(j, p);
  jac_mul_u64(q, j, BLS_X_ABS);
  jac_mul_u64(q, q, BLS_X_ABS);              // [x^2] P
  if (jac_is_inf(q)) return false;
  fp beta, bx, z2, z3, t;
  fp_load(beta, G1_BETA);
  fp_mul(bx, beta, p.x);
  fp_sqr(z2, q.z);
  fp_mul(z3, z2, q.z);
  fp_mul(t, bx, z2);
  if (!fp_eq(t, q.x)) return false;          // x(phi P) == x(-[x^2]P)
  fp_mul(t, p.y, z3);
  fp_neg(t, t);
  return fp_eq(t, q.y);                      // y(phi P) == -y([x^2]P)
}
BLS_NOINLINE bool g2_in_subgroup(const g2_aff& p) {
  if (p.inf) return true;
  g2_jac j, q, ps;
  jac_from_aff(j, p);
  jac_mul_u64(q, j, BLS_X_ABS);              // [|x|] P = -[x] P
  if (jac_is_inf(q)) return false;
  g2_psi(ps, j);                             // Z = 1
  fp2 z2, z3, t;
  fp2_sqr(z2, q.z);
  fp2_mul(z3, z2, q.z);
  fp2_mul(t, ps.x, z2);
  if (!fp2_eq(t, q.x)) return false;
  fp2_mul(t, ps.y, z3);
  fp2_neg(t, t);
  return fp2_eq(t, q.y);
}

// ---- checked decompression (ZCash encoding; `legacy` = Dash header, reference src/impls/legacy.rs:39-67,100-126,144-170)
// returns 0 ok, 7 bad encoding (DeserializationError), 8 legacy header violation (LegacyFormatError)
BLS_FN bool fp_from_be48_checked(fp& r, const uint8_t* b, uint8_t b0) {   // b0 = first byte with the flag bits cleared
  fp t;
  uint32_t ww[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const uint8_t* q = b + 4 * (11 - i);
    uint32_t hi = (i == 11) ? b0 : q[0];
    ww[i] = (hi << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  }
  fp_set_words(t, ww);
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < FP_NL; i++) c = (t.l[i] - (int32_t)FP_P[i] + c) >> FP_LB;
  if (c >= 0) return false;                  // value >= p
  fp_to_mont(r, t);
  return true;
}
BLS_FN int wire_header(uint8_t& b0, bool legacy) {
  if (legacy) {
    if (b0 != 0xc0) {
      const bool ys = (b0 & 0x80) != 0;
      b0 &= 0x7f;
      if (b0 & 0xe0) return 8;
      b0 |= 0x80;
      if (ys) b0 |= 0x20;
    }
  } else if (b0 != 0xc0 && (b0 & 0xc0) != 0x80) {
    return 7;                                // validate_modern_format, legacy.rs:71-82
  }
  return 0;
}
BLS_NOINLINE int g1_decompress(g1_jac& out, const uint8_t* b, bool legacy) {
  uint8_t b0 = b[0];
  int rc = wire_header(b0, legacy);
  if (rc) return rc;
  if (!(b0 & 0x80)) return 7;
  const bool inf = (b0 & 0x40) != 0, sign = (b0 & 0x20) != 0;
  if (inf) {
    uint32_t o = (b0 & 0x3f);
    for (int i = 1; i < 48; i++) o |= b[i];
    if (o) return 7;
    jac_set_inf(out);
    return 0;
  }
  g1_aff a;
  if (!fp_from_be48_checked(a.x, b, b0 & 0x1f)) return 7;
  fp y2, t, four;
  fp_sqr(t, a.x);
  fp_mul(t, t, a.x);
  fp_load(four, G1_B);
  fp_add(y2, t, four);
  if (!fp_sqrt(a.y, y2)) return 7;
  if (fp_lex_largest(a.y) != sign) fp_neg(a.y, a.y);
  a.inf = false;
  if (!g1_in_subgroup(a)) return 7;
  jac_from_aff(out, a);
  return 0;
}
BLS_NOINLINE int g2_decompress(g2_jac& out, const uint8_t* b, bool legacy) {
  uint8_t b0 = b[0];
  int rc = wire_header(b0, legacy);
  if (rc) return rc;
  if (!(b0 & 0x80)) return 7;
  const bool inf = (b0 & 0x40) != 0, sign = (b0 & 0x20) != 0;
  if (inf) {
    uint32_t o = (b0 & 0x3f);
    for (int i = 1; i < 96; i++) o |= b[i];
    if (o) return 7;
    jac_set_inf(out);
    return 0;
  }
  g2_aff a;
  if (!fp_from_be48_checked(a.x.c1, b, b0 & 0x1f)) return 7;
  if (!fp_from_be48_checked(a.x.c0, b + 48, b[48])) return 7;
  fp2 y2, t, bb;
  fp2_sqr(t, a.x);
  fp2_mul(t, t, a.x);
  fp2_load(bb, G2_B);
  fp2_add(y2, t, bb);
  if (!fp2_sqrt(a.y, y2)) return 7;
  if (fp2_lex_largest(a.y) != sign) fp2_neg(a.y, a.y);
  a.inf = false;
  if (!g2_in_subgroup(a)) return 7;
  jac_from_aff(out, a);
  return 0;
}
