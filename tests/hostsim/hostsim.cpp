#include <vector>
// TEST-ONLY harness: compiles the device arithmetic headers (agora-blsful_amd/csrc/*.cuh) as plain host C++ so that
// the `-m "not gpu"` suite can check every per-item device function against the oracle without a GPU.
// It is never linked into libblsgpu.so and is not a CPU fallback: the product library has no host compute path.
#include <string.h>
#include "../../agora-blsful_amd/csrc/verify.cuh"
#include "../../agora-blsful_amd/csrc/tower_split.cuh"
#include "../../agora-blsful_amd/csrc/msm2.cuh"

// callers pass blst-style Montgomery words (R = 2^384, 12 words per Fp) exactly like the RAW formats of the C ABI
static void raw_fp2(fp2& r, const uint32_t* w) { fp_from_raw(r.c0, w); fp_from_raw(r.c1, w + 12); }
static void load_g1_jac(g1_jac& p, const uint32_t* w) { fp_from_raw(p.x, w); fp_from_raw(p.y, w + 12); fp_from_raw(p.z, w + 24); }
static void load_g2_jac(g2_jac& p, const uint32_t* w) { raw_fp2(p.x, w); raw_fp2(p.y, w + 24); raw_fp2(p.z, w + 48); }
static void store_plain(uint32_t* out, const fp& a_mont) { fp t; fp_from_mont(t, a_mont); fp_get_words(out, t); }
static void store_fp12_plain(uint32_t* out, const fp12& f) {
  // w-power order: k = 2 j + i for c_i . a_j
  const fp2* c[6] = {&f.c0.a0, &f.c1.a0, &f.c0.a1, &f.c1.a1, &f.c0.a2, &f.c1.a2};
  for (int k = 0; k < 6; k++) {
    store_plain(out + 24 * k, c[k]->c0);
    store_plain(out + 24 * k + 12, c[k]->c1);
  }
}

// sum_i k_i P_i by the signed-window bucket method with window width c (no shortcuts: buckets as plain arrays)
template <class F, int E>
static void msm2_eval(jac<F>& res, int n, const F* qx, const F* qy, const uint8_t* inf, const uint64_t* subs, int words, int W) {
  const msm2_layout L = msm2_make_layout(64 * words, W);
  const size_t nb = msm2_buckets(L);
  jac<F>* bucket = new jac<F>[nb];
  for (size_t t = 0; t < nb; t++) jac_set_inf(bucket[t]);
  for (int i = 0; i < n; i++) {
    if (inf[i]) continue;
    for (int j = 0; j < E; j++) {
      uint32_t carry = 0;
      for (int w = 0; w < W; w++) {
        const int32_t d = msm2_digit(subs + ((size_t)i * E + j) * words, words, L, w, carry);
        if (!d) continue;
        F y = qy[i * E + j];
        if (d < 0) fe_neg(y, y);
        jac<F>& b = bucket[msm2_bucket_base(L, w) + (size_t)((d < 0 ? -d : d) - 1)];
        jac_madd(b, b, qx[i * E + j], y);
      }
    }
  }
  jac_set_inf(res);
  for (int w = W - 1; w >= 0; w--) {
    const int c = msm2_width(L, w);
    for (int k = 0; k < c; k++) jac_dbl(res, res);          // Horner over the windows: res = res * 2^width(w) + window sum
    jac<F> run, acc;
    jac_set_inf(run); jac_set_inf(acc);
    for (int t = (1 << (c - 1)) - 1; t >= 0; t--) { jac_add(run, run, bucket[msm2_bucket_base(L, w) + t]); jac_add(acc, acc, run); }
    jac_add(res, res, acc);
  }
  delete[] bucket;
}

extern "C" {
// 1 (what k_finalexp2s runs since round 4): the compressed squarings with one lane per Fp4 squaring and one-lane Karatsuba products
// (tower_split.cuh cyc_c_sqr overload); 0: round 3's lane-split squarings (k_finalexps, k_finalexp_seg / k_cyc_run4)
void hs_set_cyc_kara(int on) { g_cyc_kara = on; }
uint64_t hs_phase_marks[3] = {0, 0, 0};   // BLS_COUNT_FPMUL builds: the census after prepare / Miller loop / final exponentiation
int hs_device_path_only = 0;   // tools/count_fpmul.py: run exactly what the kernels run (prepare, then the lane-split pairing)
void hs_fp_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  fp x, y, z; fp_from_raw(x, a); fp_from_raw(y, b); fp_mul(z, x, y); fp_to_raw(out, z);
}
// the two conversions from the caller's words: the shifted integer with one value reduction (round 4) and the multiplication by
// 2^400 mod p (rounds 1-3).  out: the canonical limbs of each (14 + 14 words); returns 1 when they are equal
int hs_from_raw_forms(const uint32_t* a, int32_t* out) {
  fp x, y;
  fp_from_raw(x, a);
  fp_from_raw_mul(y, a);
  fp_canon(x, x);
  fp_canon(y, y);
  int same = 1;
  for (int i = 0; i < FP_NL; i++) {
    out[i] = x.l[i];
    out[FP_NL + i] = y.l[i];
    same &= x.l[i] == y.l[i];
  }
  return same;
}
// lazy limbs (any redundant form within the stated bounds) -> the caller's words, both forms; returns 1 when they are equal
int hs_to_raw_forms(const int32_t* limbs, double lb, double vb, uint32_t* out) {
  fp a;
  for (int i = 0; i < FP_NL; i++) a.l[i] = limbs[i];
  FP_TRK(a.lb = lb; a.vb = vb;)
  (void)lb; (void)vb;
  uint32_t w2[12];
  fp_to_raw(out, a);
  fp_to_raw_mul(w2, a);
  int same = 1;
  for (int i = 0; i < 12; i++) same &= out[i] == w2[i];
  return same;
}
void hs_fp_inv_var(const uint32_t* a, uint32_t* out) {   // the variable-time inversion of the lone-lane paths
  fp x, z; fp_from_raw(x, a); fp_inv_var(z, x); fp_to_raw(out, z);
}
void hs_fp_ops(const uint32_t* a, const uint32_t* b, uint32_t* out) {  // add, sub, neg, inv, sqrt-flag
  fp x, y, z; fp_from_raw(x, a); fp_from_raw(y, b);
  fp_add(z, x, y); fp_to_raw(out, z);
  fp_sub(z, x, y); fp_to_raw(out + 12, z);
  fp_neg(z, x); fp_to_raw(out + 24, z);
  fp_inv(z, x); fp_to_raw(out + 36, z);
  out[48] = fp_sqrt(z, x) ? 1 : 0; fp_to_raw(out + 49, z);
  out[61] = fp_lex_largest(x) ? 1 : 0;
}
// Lazy-limb primitives on caller-chosen limb vectors (14 signed 32-bit limbs, any redundant form within the stated
// bounds).  out: [0..13] fp_norm limbs, [14..27] fp_reduce limbs, [28..41] fp_canon limbs, [42] fp_is_zero,
// [43..54] fp_get_words(fp_canon).  lb / vb: the caller's bound claims for the tracker.
void hs_fp_lazy(const int32_t* limbs, double lb, double vb, int32_t* out) {
  fp a, t;
  for (int i = 0; i < FP_NL; i++) a.l[i] = limbs[i];
  FP_TRK(a.lb = lb; a.vb = vb;)
  (void)lb; (void)vb;
  fp_norm(t, a);
  for (int i = 0; i < FP_NL; i++) out[i] = t.l[i];
  fp_reduce(t, a);
  for (int i = 0; i < FP_NL; i++) out[14 + i] = t.l[i];
  fp_canon(t, a);
  for (int i = 0; i < FP_NL; i++) out[28 + i] = t.l[i];
  out[42] = fp_is_zero(a) ? 1 : 0;
  fp_get_words((uint32_t*)out + 43, t);
}
// fp_reduce_lin2 on lazy limbs: out = the reduced limbs of ka a + kb b
void hs_fp_reduce_lin2(const int32_t* la, double lba, double vba, int ka, const int32_t* lb_, double lbb, double vbb, int kb, int32_t* out) {
  fp a, b, t;
  for (int i = 0; i < FP_NL; i++) {
    a.l[i] = la[i];
    b.l[i] = lb_[i];
  }
  FP_TRK(a.lb = lba; a.vb = vba; b.lb = lbb; b.vb = vbb;)
  (void)lba; (void)vba; (void)lbb; (void)vbb;
  fp_reduce_lin2(t, a, ka, b, kb);
  for (int i = 0; i < FP_NL; i++) out[i] = t.l[i];
}
void hs_fp12_check(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  // a, b: fp12 as 6 fp2 in w-power order, Montgomery.  out: mul, sqr(a), inv(a), frob1(a), frob2(a) plain w-order
  fp12 x, y, z;
  fp2* cx[6] = {&x.c0.a0, &x.c1.a0, &x.c0.a1, &x.c1.a1, &x.c0.a2, &x.c1.a2};
  fp2* cy[6] = {&y.c0.a0, &y.c1.a0, &y.c0.a1, &y.c1.a1, &y.c0.a2, &y.c1.a2};
  for (int k = 0; k < 6; k++) { raw_fp2(*cx[k], a + 24 * k); raw_fp2(*cy[k], b + 24 * k); }
  fp12_mul(z, x, y); store_fp12_plain(out, z);
  fp12_sqr(z, x); store_fp12_plain(out + 144, z);
  fp12_inv(z, x); store_fp12_plain(out + 288, z);
  fp12_frob<1>(z, x); store_fp12_plain(out + 432, z);
  fp12_frob<2>(z, x); store_fp12_plain(out + 576, z);
}
void hs_hash_to_g1(const uint8_t* msg, uint32_t len, const uint8_t* dst, uint32_t dlen, uint8_t* out48) {
  g1_jac h; g1_aff a;
  hash_to_g1(h, nullptr, 0, msg, len, dst, dlen);
  jac_to_aff(a, h);
  g1_compress(out48, a, false);
}
void hs_hash_to_g2(const uint8_t* msg, uint32_t len, const uint8_t* dst, uint32_t dlen, uint8_t* out96) {
  g2_jac h; g2_aff a;
  hash_to_g2(h, nullptr, 0, msg, len, dst, dlen);
  jac_to_aff(a, h);
  g2_compress(out96, a, false);
}
// k * P for raw Jacobian inputs; compressed outputs.  also P + Q
void hs_g1_mul_add(const uint32_t* p, const uint32_t* q, const uint32_t* k, uint8_t* out_mul, uint8_t* out_add, int legacy) {
  g1_jac a, b, r; g1_aff f;
  load_g1_jac(a, p); load_g1_jac(b, q);
  jac_mul_scalar(r, a, k); jac_to_aff(f, r); g1_compress(out_mul, f, legacy);
  jac_add(r, a, b); jac_to_aff(f, r); g1_compress(out_add, f, legacy);
}
void hs_g2_mul_add(const uint32_t* p, const uint32_t* q, const uint32_t* k, uint8_t* out_mul, uint8_t* out_add) {
  g2_jac a, b, r; g2_aff f;
  load_g2_jac(a, p); load_g2_jac(b, q);
  jac_mul_scalar(r, a, k); jac_to_aff(f, r); g2_compress(out_mul, f, false);
  jac_add(r, a, b); jac_to_aff(f, r); g2_compress(out_add, f, false);
}
void hs_g2_clear_cofactor(const uint32_t* p, uint8_t* out) {
  g2_jac a, r; g2_aff f;
  load_g2_jac(a, p); g2_clear_cofactor(r, a); jac_to_aff(f, r); g2_compress(out, f, false);
}
// final_exponentiation(miller_loop(pairs)) with affine Montgomery inputs (x, y), n = 1 or 2
void hs_pairing(int n, const uint32_t* g1s, const uint32_t* g2s, uint32_t* out_plain, uint32_t* out_miller_plain) {
  g1_aff P[2]; g2_aff Q[2];
  for (int i = 0; i < n; i++) {
    fp_from_raw(P[i].x, g1s + 24 * i); fp_from_raw(P[i].y, g1s + 24 * i + 12); P[i].inf = false;
    raw_fp2(Q[i].x, g2s + 48 * i); raw_fp2(Q[i].y, g2s + 48 * i + 24); Q[i].inf = false;
  }
  fp12 f, e;
  if (n == 1) miller_loop<1>(f, P, Q); else miller_loop<2>(f, P, Q);
  if (out_miller_plain) store_fp12_plain(out_miller_plain, f);
  final_exponentiation(e, f);
  store_fp12_plain(out_plain, e);
}
// Miller loop with the precomputed -g2 line table vs the generic two-pair loop with Q1 = -g2: raw Miller values
int hs_miller_fixed_g2_matches(const uint32_t* p0, const uint32_t* q0, const uint32_t* p1) {
  g1_aff P[2]; g2_aff Q[2];
  fp_from_raw(P[0].x, p0); fp_from_raw(P[0].y, p0 + 12); P[0].inf = false;
  fp_from_raw(P[1].x, p1); fp_from_raw(P[1].y, p1 + 12); P[1].inf = false;
  raw_fp2(Q[0].x, q0); raw_fp2(Q[0].y, q0 + 24); Q[0].inf = false;
  g2_neg_gen(Q[1]);
  fp12 a, b;
  miller_loop<2>(a, P, Q);
  miller_loop_fixed_g2(b, P[0], Q[0], P[1]);
  uint32_t wa[144], wb[144];
  store_fp12_plain(wa, a); store_fp12_plain(wb, b);
  return memcmp(wa, wb, sizeof wa) == 0;
}
// the merged-line loops (normalised fixed table / two general pairs) against the plain loops: the Miller values differ by a factor
// in Fp2, the final exponentiations must be equal.  Both tower instantiations.  1 = all equal
int hs_miller_merged_matches(const uint32_t* p0, const uint32_t* q0, const uint32_t* p1, const uint32_t* q1) {
  g1_aff P[2]; g2_aff Q[2];
  fp_from_raw(P[0].x, p0); fp_from_raw(P[0].y, p0 + 12); P[0].inf = false;
  fp_from_raw(P[1].x, p1); fp_from_raw(P[1].y, p1 + 12); P[1].inf = false;
  raw_fp2(Q[0].x, q0); raw_fp2(Q[0].y, q0 + 24); Q[0].inf = false;
  raw_fp2(Q[1].x, q1); raw_fp2(Q[1].y, q1 + 24); Q[1].inf = false;
  uint32_t wa[144], wb[144];
  fp12 a, b, ea, eb;
  // two general pairs
  miller_loop<2>(a, P, Q);
  miller_loop2_merged(b, P, Q);
  final_exponentiation(ea, a); final_exponentiation(eb, b);
  store_fp12_plain(wa, ea); store_fp12_plain(wb, eb);
  if (memcmp(wa, wb, sizeof wa)) return -1;
  // fixed second argument, both tables
  for (int tb = 0; tb < 2; tb++) {
    miller_loop_fixed_g2(a, P[0], Q[0], P[1], tb ? G2NEGC_LINES : G2NEG_LINES);
    miller_loop_fixed_g2_merged(b, P[0], Q[0], P[1], tb ? G2NEGC_LINES_N : G2NEG_LINES_N);
    final_exponentiation(ea, a); final_exponentiation(eb, b);
    store_fp12_plain(wa, ea); store_fp12_plain(wb, eb);
    if (memcmp(wa, wb, sizeof wa)) return -2 - tb;
  }
  // the lane-split tower on the fixed form
  aff<hfp2> QQ; QQ.x.c[0] = Q[0].x.c0; QQ.x.c[1] = Q[0].x.c1; QQ.y.c[0] = Q[0].y.c0; QQ.y.c[1] = Q[0].y.c1; QQ.inf = false;
  fp12_t<hfp2> fs, es;
  miller_loop_fixed_g2_merged(fs, P[0], QQ, P[1], G2NEGC_LINES_N);
  final_exponentiation(es, fs);
  fp2 chk; chk.c0 = es.c0.a0.c[0]; chk.c1 = es.c0.a0.c[1];
  if (!fp2_eq(chk, eb.c0.a0)) return -4;
  chk.c0 = es.c1.a2.c[0]; chk.c1 = es.c1.a2.c[1];
  return fp2_eq(chk, eb.c1.a2) ? 1 : -5;
}
// Pairing products as the kernels of run_miller_product_tree take them (kernels.cuh k_linesp / k_line_quad / k_f12_fold4 / program
// HORNER): the plain line values of every item and entry; per entry the product over the items -- four items' values merged two by
// two and multiplied (lines_merge, fp12_from_line5, fp12_mul_by_line5_body), the quads' values folded with general products;
// then ONE Horner chain over the 68 entry products.  Lane-split tower, bound tracker on.  fixed_last: the last pair's G2 member is
// -[c] g2 and its line values come from the normalised table (k_lines_fixed).  Compared after the final exponentiation with the
// product of the plain Miller loops.  1 = equal
int hs_product_tree_matches(int n, const uint32_t* g1s, const uint32_t* g2s, int fixed_last) {
  typedef hfp2 F2;
  std::vector<g1_aff> P(n);
  std::vector<g2_aff> Q(n);
  for (int i = 0; i < n; i++) {
    fp_from_raw(P[i].x, g1s + 24 * i); fp_from_raw(P[i].y, g1s + 24 * i + 12); P[i].inf = false;
    raw_fp2(Q[i].x, g2s + 48 * i); raw_fp2(Q[i].y, g2s + 48 * i + 24); Q[i].inf = false;
  }
  // reference: one plain Miller loop per pair (the last pair's G2 member as given: the caller passes -[c] g2 there when fixed_last)
  fp12 ref, one_pair, eref;
  for (int i = 0; i < n; i++) {
    miller_loop<1>(one_pair, &P[i], &Q[i]);
    if (i == 0) ref = one_pair;
    else fp12_mul(ref, ref, one_pair);
  }
  final_exponentiation(eref, ref);
  // the items' line values
  struct line3 { F2 l0, l2, l3; };
  std::vector<std::vector<line3> > L(n, std::vector<line3>(MILLER_ENTRIES));
  for (int i = 0; i < n; i++) {
    if (fixed_last && i == n - 1) {
      for (int e = 0; e < MILLER_ENTRIES; e++) {
        F2 c;
        fp2_load(L[i][e].l0, G2NEGC_LINES_N[e]);
        fp2_load(c, G2NEGC_LINES_N[e] + 2 * FP_NL);
        fp2_mul_fp(L[i][e].l2, c, P[i].x);
        fp2_from_fp(L[i][e].l3, P[i].y);
      }
      continue;
    }
    aff<F2> QQ;
    QQ.x.c[0] = Q[i].x.c0; QQ.x.c[1] = Q[i].x.c1; QQ.y.c[0] = Q[i].y.c0; QQ.y.c[1] = Q[i].y.c1; QQ.inf = false;
    g2_hom_t<F2> T;
    T.x = QQ.x; T.y = QQ.y; fp2_one(T.z);
    for (int e = 0; e < MILLER_ENTRIES; e++) {
      if (miller_entry_is_add(e)) miller_add_step(T, L[i][e].l0, L[i][e].l2, L[i][e].l3, QQ.x, QQ.y, P[i].x, P[i].y);
      else miller_dbl_step(T, L[i][e].l0, L[i][e].l2, L[i][e].l3, P[i].x, P[i].y);
    }
  }
  // per entry: quads, then a fold by fours
  line3 neutral;
  fp2_one(neutral.l0); fp2_zero(neutral.l2); fp2_zero(neutral.l3);
  const int q = (n + 3) / 4;
  fp12_t<F2> F;
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    std::vector<fp12_t<F2> > v(q);
    for (int j = 0; j < q; j++) {
      const line3& a = L[j][e];
      const line3& b = j + q < n ? L[j + q][e] : neutral;
      const line3& c = j + 2 * q < n ? L[j + 2 * q][e] : neutral;
      const line3& d = j + 3 * q < n ? L[j + 3 * q][e] : neutral;
      line5_t<F2> M;
      lines_merge(M, a.l0, a.l2, a.l3, b.l0, b.l2, b.l3);
      fp12_from_line5(v[j], M);
      fp12_reduce(v[j], v[j]);
      lines_merge(M, c.l0, c.l2, c.l3, d.l0, d.l2, d.l3);
      fp12_mul_by_line5_body(v[j], M);
    }
    int m = q;
    while (m > 1) {
      const int mo = (m + 3) / 4;
      for (int j = 0; j < mo; j++)
        for (int k = 1; k < 4; k++)
          if (j + k * mo < m) fp12_mul(v[j], v[j], v[j + k * mo]);
      m = mo;
    }
    if (e == 0) {
      F = v[0];
    } else {
      if (!miller_entry_is_add(e)) fp12_sqr(F, F);
      fp12_mul(F, F, v[0]);
    }
  }
  fp12_t<F2> Fc, es;
  fp12_conj(Fc, F);
  final_exponentiation(es, Fc);
  fp2 chk;
  const F2* got[6] = {&es.c0.a0, &es.c0.a1, &es.c0.a2, &es.c1.a0, &es.c1.a1, &es.c1.a2};
  const fp2* want[6] = {&eref.c0.a0, &eref.c0.a1, &eref.c0.a2, &eref.c1.a0, &eref.c1.a1, &eref.c1.a2};
  for (int k = 0; k < 6; k++) {
    chk.c0 = got[k]->c[0];
    chk.c1 = got[k]->c[1];
    if (!fp2_eq(chk, *want[k])) return -1 - k;
  }
  return 1;
}
// cyclotomic squaring vs generic squaring on an element of the cyclotomic subgroup
int hs_cyclotomic_check(int n, const uint32_t* g1s, const uint32_t* g2s) {
  g1_aff P[1]; g2_aff Q[1];
  fp_from_raw(P[0].x, g1s); fp_from_raw(P[0].y, g1s + 12); P[0].inf = false;
  raw_fp2(Q[0].x, g2s); raw_fp2(Q[0].y, g2s + 24); Q[0].inf = false;
  fp12 f, e, a, b;
  miller_loop<1>(f, P, Q);
  final_exponentiation(e, f);
  fp12_cyclotomic_sqr(a, e);
  fp12_sqr(b, e);
  uint32_t wa[144], wb[144];
  store_fp12_plain(wa, a); store_fp12_plain(wb, b);
  (void)n;
  return memcmp(wa, wb, sizeof wa) == 0;
}
// the lane-split point arithmetic (jac<hfp2>, host emulation) against the one-lane code: P + Q, 2P, clear_cofactor(P)
static void split_jac(jac<hfp2>& r, const g2_jac& p) {
  r.x.c[0] = p.x.c0; r.x.c[1] = p.x.c1; r.y.c[0] = p.y.c0; r.y.c[1] = p.y.c1; r.z.c[0] = p.z.c0; r.z.c[1] = p.z.c1;
}
static void unsplit_jac(g2_jac& r, const jac<hfp2>& p) {
  r.x.c0 = p.x.c[0]; r.x.c1 = p.x.c[1]; r.y.c0 = p.y.c[0]; r.y.c1 = p.y.c[1]; r.z.c0 = p.z.c[0]; r.z.c1 = p.z.c[1];
}
void hs_g2_split_ops(const uint32_t* p, const uint32_t* q, uint8_t* out_add, uint8_t* out_dbl, uint8_t* out_clear) {
  g2_jac a, b, r; g2_aff f;
  load_g2_jac(a, p); load_g2_jac(b, q);
  jac<hfp2> sa, sb, sr;
  split_jac(sa, a); split_jac(sb, b);
  jac_add(sr, sa, sb); unsplit_jac(r, sr); jac_to_aff(f, r); g2_compress(out_add, f, false);
  jac_dbl(sr, sa); unsplit_jac(r, sr); jac_to_aff(f, r); g2_compress(out_dbl, f, false);
  g2_clear_cofactor(sr, sa); unsplit_jac(r, sr); jac_to_aff(f, r); g2_compress(out_clear, f, false);
}
// a^x by compressed squarings vs the plain Granger-Scott chain, on the cyclotomic element of a pairing; both towers
int hs_pow_x_compressed_check(const uint32_t* g1s, const uint32_t* g2s) {
  g1_aff P[1]; g2_aff Q[1];
  fp_from_raw(P[0].x, g1s); fp_from_raw(P[0].y, g1s + 12); P[0].inf = false;
  raw_fp2(Q[0].x, g2s); raw_fp2(Q[0].y, g2s + 24); Q[0].inf = false;
  fp12 f, e, a, b;
  miller_loop<1>(f, P, Q);
  final_exponentiation(e, f);
  fp12_pow_x_plain(a, e);
  if (!fp12_pow_x_compressed(b, e)) return -1;
  uint32_t wa[144], wb[144];
  store_fp12_plain(wa, a); store_fp12_plain(wb, b);
  if (memcmp(wa, wb, sizeof wa)) return 0;
  if (!fp12_pow_x_compressed4(b, e)) return -6;       // the four-power chain with the plain tail (k_finalexp2s), one-lane tower
  store_fp12_plain(wb, b);
  if (memcmp(wa, wb, sizeof wa)) return -7;
  // the lane-split instantiation
  fp12_t<hfp2> es, as, bs;
  const fp2* src[6] = {&e.c0.a0, &e.c0.a1, &e.c0.a2, &e.c1.a0, &e.c1.a1, &e.c1.a2};
  hfp2* dst[6] = {&es.c0.a0, &es.c0.a1, &es.c0.a2, &es.c1.a0, &es.c1.a1, &es.c1.a2};
  for (int k = 0; k < 6; k++) { dst[k]->c[0] = src[k]->c0; dst[k]->c[1] = src[k]->c1; }
  fp12_pow_x_plain(as, es);
  if (!fp12_pow_x_compressed(bs, es)) return -2;
  const hfp2* ra[6] = {&as.c0.a0, &as.c0.a1, &as.c0.a2, &as.c1.a0, &as.c1.a1, &as.c1.a2};
  const hfp2* rb[6] = {&bs.c0.a0, &bs.c0.a1, &bs.c0.a2, &bs.c1.a0, &bs.c1.a1, &bs.c1.a2};
  for (int k = 0; k < 6; k++)
    if (!fp2_eq(*ra[k], *rb[k])) return -3;
  // and against the one-lane result
  fp2 chk; chk.c0 = as.c0.a0.c[0]; chk.c1 = as.c0.a0.c[1];
  return fp2_eq(chk, a.c0.a0) ? 1 : -4;
}
// the compressed squaring with one lane per Fp4 squaring against the lane-split form and the one-lane tower, `steps` squarings in a row
// on the cyclotomic element of a pairing: the canonical coordinates must agree after every step.  1 = all equal
int hs_cyc_kara_check(const uint32_t* g1s, const uint32_t* g2s, int steps) {
  g1_aff P[1]; g2_aff Q[1];
  fp_from_raw(P[0].x, g1s); fp_from_raw(P[0].y, g1s + 12); P[0].inf = false;
  raw_fp2(Q[0].x, g2s); raw_fp2(Q[0].y, g2s + 24); Q[0].inf = false;
  fp12 f, e, er;
  miller_loop<1>(f, P, Q);
  final_exponentiation(e, f);
  fp12_reduce(er, e);
  cyc_c<fp2> c1;
  cyc_compress(c1, er);
  cyc_c<hfp2> ck, cs;
  const fp2* src[4] = {&c1.z2, &c1.z3, &c1.z4, &c1.z5};
  hfp2* dk[4] = {&ck.z2, &ck.z3, &ck.z4, &ck.z5};
  hfp2* ds[4] = {&cs.z2, &cs.z3, &cs.z4, &cs.z5};
  for (int k = 0; k < 4; k++) { dk[k]->c[0] = src[k]->c0; dk[k]->c[1] = src[k]->c1; *ds[k] = *dk[k]; }
  const int keep = g_cyc_kara;
  int ok = 1;
  for (int i = 0; i < steps && ok == 1; i++) {
    cyc_c_sqr(c1, c1);
    g_cyc_kara = 1; cyc_c_sqr(ck, ck);
    g_cyc_kara = 0; cyc_c_sqr(cs, cs);
    const fp2* w[4] = {&c1.z2, &c1.z3, &c1.z4, &c1.z5};
    for (int k = 0; k < 4; k++) {
      if (!fp2_eq(*dk[k], *ds[k])) ok = -1 - i;
      fp2 chk; chk.c0 = dk[k]->c[0]; chk.c1 = dk[k]->c[1];
      if (!fp2_eq(chk, *w[k])) ok = -1000 - i;
    }
  }
  g_cyc_kara = keep;
  return ok;
}
// one Fp2 product by the one-lane Karatsuba pass and (a - b)(a - xi b) by its difference form, on caller-chosen reduced operands
// (Montgomery words): out = the two results as Montgomery words (24 + 24)
void hs_fp2_kara(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  fp a0, a1, b0, b1, r0, r1;
  fp_from_raw(a0, a); fp_from_raw(a1, a + 12); fp_from_raw(b0, b); fp_from_raw(b1, b + 12);
  fp2_mul_kara(r0, r1, a0, a1, b0, b1);
  fp_to_raw(out, r0); fp_to_raw(out + 12, r1);
  fp2_mul_kara_diffs(r0, r1, a0, a1, b0, b1);
  fp_to_raw(out + 24, r0); fp_to_raw(out + 36, r1);
}
// a^x for a = 1: every compressed coordinate vanishes, the compressed chains must decline (z2 = 0: the other decompression formula)
// and fp12_pow_x must come back with 1 through the plain chain -- the fallback of k_finalexp2s / fx_pow_plain.  Both towers, both
// chain forms.  1 = as expected
int hs_pow_x_identity_check(void) {
  fp12 one, r;
  fp12_one(one);
  if (fp12_pow_x_compressed(r, one) || fp12_pow_x_compressed4(r, one)) return -1;
  fp12_pow_x(r, one);
  if (!fp12_is_one(r)) return -2;
  fp12_t<hfp2> os, rs;
  fp12_one(os);
  const int keep = g_cyc_kara;
  int ok = 1;
  for (int on = 0; on < 2 && ok == 1; on++) {
    g_cyc_kara = on;
    if (fp12_pow_x_compressed(rs, os)) ok = -3 - on;
    fp12_pow_x(rs, os);
    if (!fp12_is_one(rs)) ok = -5 - on;
  }
  g_cyc_kara = keep;
  return ok;
}
// checked decompression: returns the status code; on success writes the re-compressed (modern) bytes
int hs_decompress(int group, const uint8_t* in, int legacy, uint8_t* out) {
  if (group == 1) {
    g1_jac p; int rc = g1_decompress(p, in, legacy != 0); if (rc) return rc;
    g1_aff a; jac_to_aff(a, p); g1_compress(out, a, false); return 0;
  }
  g2_jac p; int rc = g2_decompress(p, in, legacy != 0); if (rc) return rc;
  g2_aff a; jac_to_aff(a, p); g2_compress(out, a, false); return 0;
}
// the lane-split tower (tower_split.cuh host emulation) on the same pairs: what k_miller2s / k_finalexps compute
static void to_split(aff<hfp2>& r, const g2_aff& q) { r.x.c[0] = q.x.c0; r.x.c[1] = q.x.c1; r.y.c[0] = q.y.c0; r.y.c[1] = q.y.c1; r.inf = q.inf; }
// 1 (what the kernels do for Bls12381G1Impl): the message point stays uncleared, pair 1 is (sig, -[c] g2) with its own line table
// (csrc/g2neg_lines.cuh); 0: the textbook form with the cleared hash and -g2
int hs_no_clear = 1;
// 1 (what the lane-split batch kernels do since round 3): the two-kernel Miller loop over merged line values; 0: the one-kernel loop
int hs_merged_lines = 1;
static int verdict_split(int sig_group, const g1_aff* P, const g2_aff* Q) {
  aff<hfp2> QQ[2];
  to_split(QQ[0], Q[0]); to_split(QQ[1], Q[1]);
  fp12_t<hfp2> f;
  if (hs_merged_lines) {   // what k_lines2s + k_millerf2s compute: the two line values of a step merged before they touch f
    if (sig_group == 1) miller_loop_fixed_g2_merged(f, P[0], QQ[0], P[1], hs_no_clear ? G2NEGC_LINES_N : G2NEG_LINES_N);
    else miller_loop2_merged(f, P, QQ);
  } else {
    if (sig_group == 1) miller_loop_fixed_g2(f, P[0], QQ[0], P[1], hs_no_clear ? G2NEGC_LINES : G2NEG_LINES);
    else miller_loop<2>(f, P, QQ);
  }
#ifdef BLS_COUNT_FPMUL
  hs_phase_marks[1] = g_fpmul_halves;
#endif
  const int v = pairing_verdict(f);
#ifdef BLS_COUNT_FPMUL
  hs_phase_marks[2] = g_fpmul_halves;
#endif
  return v;
}
int hs_verify(int sig_group, const uint32_t* pk, const uint32_t* sig, int aug, const uint8_t* msg, uint32_t len,
              const uint8_t* dst, uint32_t dlen) {
  g1_aff P[2]; g2_aff Q[2];
  int st;
  if (sig_group == 1) {
    g2_jac k; g1_jac s; load_g2_jac(k, pk); load_g1_jac(s, sig);
    st = prepare_g1impl(P, Q, k, s, aug, msg, len, dst, dlen, -1, hs_no_clear != 0);
  } else {
    g1_jac k; g2_jac s; load_g1_jac(k, pk); load_g2_jac(s, sig);
    st = prepare_g2impl(P, Q, k, s, aug, msg, len, dst, dlen);
  }
  if (st != BLS_OK) return st;
#ifdef BLS_COUNT_FPMUL
  hs_phase_marks[0] = g_fpmul_halves;
#endif
  if (hs_device_path_only) return verdict_split(sig_group, P, Q);
  fp12 f;
  if (sig_group == 1) miller_loop_fixed_g2(f, P[0], Q[0], P[1], hs_no_clear ? G2NEGC_LINES : G2NEG_LINES);   // as k_miller2 does for Bls12381G1Impl
  else miller_loop<2>(f, P, Q);
  const int v = pairing_verdict(f);
  const int vs = verdict_split(sig_group, P, Q);
  return v == vs ? v : -100 - vs;    // the two tower instantiations must agree
}

// ---- MSM second generation (msm2.cuh): scalar decomposition, point images, mixed addition, and a straightforward host
// evaluation of the signed-window bucket method built from exactly the per-item functions the kernels use
void hs_msm2_decompose(int group, const uint32_t* k, uint64_t* a) {
  if (group == 1) msm2_decompose_g1(a, k); else msm2_decompose_g2(a, k);
}
// out: group 1: 2 x 48 bytes, group 2: 4 x 96 bytes (the compressed images Q_j)
void hs_msm2_images(int group, const uint32_t* p, uint8_t* out) {
  if (group == 1) {
    g1_jac a; g1_aff f, q; load_g1_jac(a, p); jac_to_aff(f, a);
    fp qx[2], qy[2]; msm2_images_g1(qx, qy, f);
    for (int j = 0; j < 2; j++) { q.x = qx[j]; q.y = qy[j]; q.inf = false; g1_compress(out + 48 * j, q, false); }
  } else {
    g2_jac a; g2_aff f, q; load_g2_jac(a, p); jac_to_aff(f, a);
    fp2 qx[4], qy[4]; msm2_images_g2(qx, qy, f);
    for (int j = 0; j < 4; j++) { q.x = qx[j]; q.y = qy[j]; q.inf = false; g2_compress(out + 96 * j, q, false); }
  }
}
// acc (raw Jacobian, may be the identity) + affine image of q; also through the lane-split tower for G2
void hs_msm2_madd(int group, const uint32_t* acc, const uint32_t* q, int negate, uint8_t* out, uint8_t* out_split) {
  if (group == 1) {
    g1_jac a, b, r; g1_aff f; load_g1_jac(a, acc); load_g1_jac(b, q); jac_to_aff(f, b);
    fp x, y; fp_reduce(x, f.x); fp_reduce(y, f.y);
    if (negate) fp_neg(y, y);
    jac_madd(r, a, x, y); jac_to_aff(f, r); g1_compress(out, f, false);
  } else {
    g2_jac a, b, r; g2_aff f; load_g2_jac(a, acc); load_g2_jac(b, q); jac_to_aff(f, b);
    fp2 x, y; fp2_reduce(x, f.x); fp2_reduce(y, f.y);
    if (negate) fp2_neg(y, y);
    jac_madd(r, a, x, y);
    g2_aff g; jac_to_aff(g, r); g2_compress(out, g, false);
    jac<hfp2> sa, sr; split_jac(sa, a);
    hfp2 sx, sy; sx.c[0] = x.c0; sx.c[1] = x.c1; sy.c[0] = y.c0; sy.c[1] = y.c1;
    jac_madd(sr, sa, sx, sy); unsplit_jac(r, sr); jac_to_aff(g, r); g2_compress(out_split, g, false);
  }
}
void hs_msm2_small(int group, int n, const uint32_t* pts, const uint32_t* scalars, int W, uint8_t* out) {
  uint8_t* inf = new uint8_t[n];
  if (group == 1) {
    fp* qx = new fp[2 * n]; fp* qy = new fp[2 * n]; uint64_t* subs = new uint64_t[4 * n];
    for (int i = 0; i < n; i++) {
      g1_jac a; g1_aff f; load_g1_jac(a, pts + 36 * i);
      inf[i] = jac_is_inf(a);
      if (!inf[i]) { jac_to_aff(f, a); msm2_images_g1(qx + 2 * i, qy + 2 * i, f); }
      msm2_decompose_g1(subs + 4 * i, scalars + 8 * i);
    }
    g1_jac r; msm2_eval<fp, 2>(r, n, qx, qy, inf, subs, 2, W);
    g1_aff f; jac_to_aff(f, r); g1_compress(out, f, false);
    delete[] qx; delete[] qy; delete[] subs;
  } else {
    fp2* qx = new fp2[4 * n]; fp2* qy = new fp2[4 * n]; uint64_t* subs = new uint64_t[4 * n];
    for (int i = 0; i < n; i++) {
      g2_jac a; g2_aff f; load_g2_jac(a, pts + 72 * i);
      inf[i] = jac_is_inf(a);
      if (!inf[i]) { jac_to_aff(f, a); msm2_images_g2(qx + 4 * i, qy + 4 * i, f); }
      msm2_decompose_g2(subs + 4 * i, scalars + 8 * i);
    }
    g2_jac r; msm2_eval<fp2, 4>(r, n, qx, qy, inf, subs, 1, W);
    g2_aff f; jac_to_aff(f, r); g2_compress(out, f, false);
    delete[] qx; delete[] qy; delete[] subs;
  }
  delete[] inf;
}
}
