/* bls381_oracle.c -- plain-C CPU restatement of the reference's verify path.  ORACLE: test infrastructure only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this (liboracle.so); the product
 * library never links or calls it.  It restates, independently of the HIP code (6 x 64-bit limbs with __int128,
 * Jacobian twist coordinates in the Miller loop, inversion-based SSWU, affine isogeny evaluation):
 *   arithmetic delegated by the reference to the un-vendored blstrs_plus 0.8 / blst (Cargo.toml:21,23,28), from the
 *     published specs: BLS12-381, RFC 9380 BLS12381G{1,2}_XMD:SHA-256_SSWU_RO_, ZCash compressed encoding;
 *   control flow of core_verify (reference src/traits/sig_core.rs:120-146), the scheme wrappers
 *     (src/traits/sig_basic.rs:36-38, sig_aug.rs:20-24, sig_pop.rs:37-39), pairing argument order
 *     (src/helpers.rs:41-63), and verify_secure (src/secure_aggregation.rs:37-106,173-208).
 * Pinned by the reference's known-answer vectors through tests/test_oracle_c.py (K2: C++ signatures verify; K4: the
 * 57-signer production vector) and cross-checked against oracle/py on random inputs.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
/* allocation and thread helpers of the threaded CPU legs: a failed malloc ends the process with a message (this is test
 * infrastructure: there is nothing to fall back to), a failed pthread_create runs that job on the calling thread instead of
 * leaving an uninitialised pthread_t for pthread_join (advisor finding, round 2) */
static void* xmalloc(size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p) { fprintf(stderr, "oracle/c: out of memory (%zu bytes)\n", bytes); abort(); }
  return p;
}
typedef struct { pthread_t th; int started; } bo_thread;
static void bo_spawn(bo_thread* t, void* (*fn)(void*), void* arg, int inline_only) {
  t->started = 0;
  if (!inline_only && pthread_create(&t->th, 0, fn, arg) == 0) { t->started = 1; return; }
  fn(arg);
}
static void bo_join(bo_thread* t) { if (t->started) pthread_join(t->th, 0); }
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "consts.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 a0, a1, a2; } fp6;
typedef struct { fp6 c0, c1; } fp12;

static fp FP_ZERO, FP_ONE_M, FP_R2;
static int g_init = 0;

/* ------------------------------------------------------------------ Fp */
static int fp_is_zero(const fp* a) { uint64_t o = 0; for (int i = 0; i < 6; i++) o |= a->l[i]; return o == 0; }
static int fp_eq(const fp* a, const fp* b) { uint64_t o = 0; for (int i = 0; i < 6; i++) o |= a->l[i] ^ b->l[i]; return o == 0; }
static int ge_p(const uint64_t* t) {
  for (int i = 5; i >= 0; i--) { if (t[i] > K_P[i]) return 1; if (t[i] < K_P[i]) return 0; }
  return 1;
}
static void sub_p(uint64_t* t) {
  u128 bw = 0;
  for (int i = 0; i < 6; i++) { u128 d = (u128)t[i] - K_P[i] - bw; t[i] = (uint64_t)d; bw = (d >> 64) & 1; }
}
static fp fp_add(fp a, fp b) {
  fp r; u128 c = 0;
  for (int i = 0; i < 6; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  if (ge_p(r.l)) sub_p(r.l);
  return r;
}
static fp fp_sub(fp a, fp b) {
  fp r; u128 bw = 0;
  for (int i = 0; i < 6; i++) { u128 d = (u128)a.l[i] - b.l[i] - bw; r.l[i] = (uint64_t)d; bw = (d >> 64) & 1; }
  if (bw) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)r.l[i] + K_P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
  return r;
}
static fp fp_neg(fp a) { return fp_is_zero(&a) ? a : fp_sub(FP_ZERO, a); }
static fp fp_mul(fp a, fp b) {           /* Montgomery product, operand scanning */
  uint64_t t[8] = {0};
  for (int i = 0; i < 6; i++) {
    u128 c = 0;
    for (int j = 0; j < 6; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * K_N0;
    c = (u128)m * K_P[0] + t[0]; c >>= 64;
    for (int j = 1; j < 6; j++) { c += (u128)m * K_P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
  }
  fp r; memcpy(r.l, t, 48);
  if (t[6] || ge_p(r.l)) sub_p(r.l);
  return r;
}
static fp fp_sqr(fp a) { return fp_mul(a, a); }
static fp fp_from_plain(const uint64_t* w) { fp t; memcpy(t.l, w, 48); return fp_mul(t, FP_R2); }
static fp fp_to_plain(fp a) { fp one = FP_ZERO; one.l[0] = 1; return fp_mul(a, one); }
static fp fp_pow(fp a, const uint64_t* e, int bits) {
  fp r = FP_ONE_M;
  for (int i = bits - 1; i >= 0; i--) { r = fp_sqr(r); if ((e[i >> 6] >> (i & 63)) & 1) r = fp_mul(r, a); }
  return r;
}
static fp fp_inv(fp a) { return fp_pow(a, K_EXP_INV, K_EXP_INV_BITS); }
static int fp_is_square(fp a) { if (fp_is_zero(&a)) return 1; fp t = fp_pow(a, K_EXP_LEG, K_EXP_LEG_BITS); return fp_eq(&t, &FP_ONE_M); }
static int fp_sqrt(fp* r, fp a) { fp s = fp_pow(a, K_EXP_SQRT, K_EXP_SQRT_BITS); fp c = fp_sqr(s); *r = s; return fp_eq(&c, &a); }
static int fp_parity(fp a) { return (int)(fp_to_plain(a).l[0] & 1); }
static int fp_lex_largest(fp a) {        /* plain value > (p-1)/2 */
  fp t = fp_to_plain(a);
  for (int i = 5; i >= 0; i--) { if (t.l[i] > K_HALF_P[i]) return 1; if (t.l[i] < K_HALF_P[i]) return 0; }
  return 0;
}
static fp fp_from_be(const uint8_t* b, int n) {   /* n <= 48 big-endian bytes, value < 2^384, reduced via Montgomery */
  uint64_t w[6] = {0};
  for (int i = 0; i < n; i++) w[(n - 1 - i) / 8] |= (uint64_t)b[i] << (8 * ((n - 1 - i) % 8));
  return fp_from_plain(w);              /* fp_mul accepts an unreduced first operand < 2^384 */
}
static void fp_to_be48(uint8_t* out, fp a) {
  fp t = fp_to_plain(a);
  for (int i = 0; i < 48; i++) out[i] = (uint8_t)(t.l[(47 - i) / 8] >> (8 * ((47 - i) % 8)));
}

/* ------------------------------------------------------------------ Fp2 */
static fp2 FP2_ZERO, FP2_ONE;
static fp2 f2_add(fp2 a, fp2 b) { fp2 r = {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; return r; }
static fp2 f2_sub(fp2 a, fp2 b) { fp2 r = {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; return r; }
static fp2 f2_neg(fp2 a) { fp2 r = {fp_neg(a.c0), fp_neg(a.c1)}; return r; }
static fp2 f2_dbl(fp2 a) { return f2_add(a, a); }
static fp2 f2_conj(fp2 a) { fp2 r = {a.c0, fp_neg(a.c1)}; return r; }
static fp2 f2_mul(fp2 a, fp2 b) {
  fp t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
  fp m = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
  fp2 r = {fp_sub(t0, t1), fp_sub(fp_sub(m, t0), t1)};
  return r;
}
static fp2 f2_sqr(fp2 a) {
  fp m = fp_mul(a.c0, a.c1);
  fp2 r = {fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1)), fp_add(m, m)};
  return r;
}
static fp2 f2_mul_fp(fp2 a, fp k) { fp2 r = {fp_mul(a.c0, k), fp_mul(a.c1, k)}; return r; }
static fp2 f2_mul_xi(fp2 a) { fp2 r = {fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; return r; }
static fp f2_norm(fp2 a) { return fp_add(fp_sqr(a.c0), fp_sqr(a.c1)); }
static fp2 f2_inv(fp2 a) { fp n = fp_inv(f2_norm(a)); fp2 r = {fp_mul(a.c0, n), fp_neg(fp_mul(a.c1, n))}; return r; }
static int f2_is_zero(fp2 a) { return fp_is_zero(&a.c0) && fp_is_zero(&a.c1); }
static int f2_eq(fp2 a, fp2 b) { return fp_eq(&a.c0, &b.c0) && fp_eq(&a.c1, &b.c1); }
static int f2_is_square(fp2 a) { return fp_is_square(f2_norm(a)); }
static fp2 f2_from_plain(const uint64_t* w) { fp2 r = {fp_from_plain(w), fp_from_plain(w + 6)}; return r; }
static int f2_sqrt(fp2* r, fp2 a) {      /* complex method */
  if (f2_is_zero(a)) { *r = FP2_ZERO; return 1; }
  fp s;
  if (fp_is_zero(&a.c1)) {
    if (fp_sqrt(&s, a.c0)) { r->c0 = s; r->c1 = FP_ZERO; return 1; }
    int ok = fp_sqrt(&s, fp_neg(a.c0)); r->c0 = FP_ZERO; r->c1 = s; return ok;
  }
  fp n;
  if (!fp_sqrt(&n, f2_norm(a))) return 0;
  fp half = fp_inv(fp_add(FP_ONE_M, FP_ONE_M));
  if (!fp_sqrt(&s, fp_mul(fp_add(a.c0, n), half)) && !fp_sqrt(&s, fp_mul(fp_sub(a.c0, n), half))) return 0;
  r->c0 = s; r->c1 = fp_mul(a.c1, fp_inv(fp_add(s, s)));
  return f2_eq(f2_sqr(*r), a);
}
static int f2_sgn0(fp2 a) {
  fp t0 = fp_to_plain(a.c0), t1 = fp_to_plain(a.c1);
  int s0 = (int)(t0.l[0] & 1), z0 = fp_is_zero(&t0), s1 = (int)(t1.l[0] & 1);
  return s0 | (z0 & s1);
}
static int f2_lex_largest(fp2 a) { return fp_is_zero(&a.c1) ? fp_lex_largest(a.c0) : fp_lex_largest(a.c1); }

/* ------------------------------------------------------------------ Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v) */
static fp6 f6_add(fp6 a, fp6 b) { fp6 r = {f2_add(a.a0, b.a0), f2_add(a.a1, b.a1), f2_add(a.a2, b.a2)}; return r; }
static fp6 f6_sub(fp6 a, fp6 b) { fp6 r = {f2_sub(a.a0, b.a0), f2_sub(a.a1, b.a1), f2_sub(a.a2, b.a2)}; return r; }
static fp6 f6_neg(fp6 a) { fp6 r = {f2_neg(a.a0), f2_neg(a.a1), f2_neg(a.a2)}; return r; }
static fp6 f6_mul_v(fp6 a) { fp6 r = {f2_mul_xi(a.a2), a.a0, a.a1}; return r; }
static fp6 f6_mul(fp6 a, fp6 b) {
  fp2 v0 = f2_mul(a.a0, b.a0), v1 = f2_mul(a.a1, b.a1), v2 = f2_mul(a.a2, b.a2);
  fp6 r;
  r.a0 = f2_add(v0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a.a1, a.a2), f2_add(b.a1, b.a2)), v1), v2)));
  r.a1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.a0, a.a1), f2_add(b.a0, b.a1)), v0), v1), f2_mul_xi(v2));
  r.a2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.a0, a.a2), f2_add(b.a0, b.a2)), v0), v2), v1);
  return r;
}
static fp6 f6_inv(fp6 a) {
  fp2 t0 = f2_sub(f2_sqr(a.a0), f2_mul_xi(f2_mul(a.a1, a.a2)));
  fp2 t1 = f2_sub(f2_mul_xi(f2_sqr(a.a2)), f2_mul(a.a0, a.a1));
  fp2 t2 = f2_sub(f2_sqr(a.a1), f2_mul(a.a0, a.a2));
  fp2 d = f2_add(f2_mul(a.a0, t0), f2_mul_xi(f2_add(f2_mul(a.a2, t1), f2_mul(a.a1, t2))));
  fp2 di = f2_inv(d);
  fp6 r = {f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)};
  return r;
}
static fp12 F12_ONE;
static fp12 f12_mul(fp12 a, fp12 b) {
  fp6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
  fp12 r;
  r.c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), t0), t1);
  r.c0 = f6_add(t0, f6_mul_v(t1));
  return r;
}
static fp12 f12_sqr(fp12 a) {
  fp6 t = f6_mul(a.c0, a.c1);
  fp12 r;
  r.c0 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_v(a.c1))), t), f6_mul_v(t));
  r.c1 = f6_add(t, t);
  return r;
}
static fp12 f12_conj(fp12 a) { fp12 r = {a.c0, f6_neg(a.c1)}; return r; }
static fp12 f12_inv(fp12 a) {
  fp6 t = f6_inv(f6_sub(f6_mul(a.c0, a.c0), f6_mul_v(f6_mul(a.c1, a.c1))));
  fp12 r = {f6_mul(a.c0, t), f6_neg(f6_mul(a.c1, t))};
  return r;
}
static fp2 FROB[2][6];
static fp12 f12_frob(fp12 a, int j) {    /* a^(p^j), j = 1, 2: coefficient of w^k = c_{k&1}.a_{k>>1} */
  fp2* c[6] = {&a.c0.a0, &a.c1.a0, &a.c0.a1, &a.c1.a1, &a.c0.a2, &a.c1.a2};
  for (int k = 0; k < 6; k++) {
    fp2 v = (j == 1) ? f2_conj(*c[k]) : *c[k];
    *c[k] = k ? f2_mul(v, FROB[j - 1][k]) : v;
  }
  return a;
}
static int f12_is_one(fp12 a) {
  return f2_eq(a.c0.a0, FP2_ONE) && f2_is_zero(a.c0.a1) && f2_is_zero(a.c0.a2) && f2_is_zero(a.c1.a0) && f2_is_zero(a.c1.a1) && f2_is_zero(a.c1.a2);
}
/* squaring in the cyclotomic subgroup (Granger-Scott), via three Fp4 squarings */
static void f4_sqr(fp2* r0, fp2* r1, fp2 a, fp2 b) {
  fp2 t0 = f2_sqr(a), t1 = f2_sqr(b);
  *r0 = f2_add(f2_mul_xi(t1), t0);
  *r1 = f2_sub(f2_sub(f2_sqr(f2_add(a, b)), t0), t1);
}
static fp12 f12_cyc_sqr(fp12 f) {
  fp2 z0 = f.c0.a0, z4 = f.c0.a1, z3 = f.c0.a2, z2 = f.c1.a0, z1 = f.c1.a1, z5 = f.c1.a2, t0, t1, t2, t3;
  f4_sqr(&t0, &t1, z0, z1);
  z0 = f2_add(f2_dbl(f2_sub(t0, z0)), t0);
  z1 = f2_add(f2_dbl(f2_add(t1, z1)), t1);
  f4_sqr(&t0, &t1, z2, z3);
  f4_sqr(&t2, &t3, z4, z5);
  z4 = f2_add(f2_dbl(f2_sub(t0, z4)), t0);
  z5 = f2_add(f2_dbl(f2_add(t1, z5)), t1);
  t0 = f2_mul_xi(t3);
  z2 = f2_add(f2_dbl(f2_add(t0, z2)), t0);
  z3 = f2_add(f2_dbl(f2_sub(t2, z3)), t2);
  fp12 r = {{z0, z4, z3}, {z2, z1, z5}};
  return r;
}
/* f * (l0 + l2 w^2 + l3 w^3) */
static fp12 f12_mul_line(fp12 f, fp2 l0, fp2 l2, fp2 l3) {
  fp12 l = {{l0, l2, FP2_ZERO}, {FP2_ZERO, l3, FP2_ZERO}};
  /* sparse operand: skip the products with structural zeros */
  fp6 t0, t1, s = f6_add(f.c0, f.c1);
  fp2 l23 = f2_add(l2, l3);
  t0.a0 = f2_add(f2_mul(f.c0.a0, l0), f2_mul_xi(f2_mul(f.c0.a2, l2)));
  t0.a1 = f2_add(f2_mul(f.c0.a0, l2), f2_mul(f.c0.a1, l0));
  t0.a2 = f2_add(f2_mul(f.c0.a2, l0), f2_mul(f.c0.a1, l2));
  t1.a0 = f2_mul_xi(f2_mul(f.c1.a2, l3));
  t1.a1 = f2_mul(f.c1.a0, l3);
  t1.a2 = f2_mul(f.c1.a1, l3);
  fp6 m;
  m.a0 = f2_add(f2_mul(s.a0, l0), f2_mul_xi(f2_mul(s.a2, l23)));
  m.a1 = f2_add(f2_mul(s.a0, l23), f2_mul(s.a1, l0));
  m.a2 = f2_add(f2_mul(s.a2, l0), f2_mul(s.a1, l23));
  (void)l;
  fp12 r = {f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(m, t0), t1)};
  return r;
}

/* ------------------------------------------------------------------ curves, Jacobian; Z == 0 is infinity */
typedef struct { fp x, y, z; } g1p;
typedef struct { fp2 x, y, z; } g2p;
#define CURVE_TEMPLATE(G, F, ADD, SUB, MUL, SQR, DBL, NEG, ISZ, EQ)                                              \
  static int G##_is_inf(const G* p) { return ISZ(p->z); }                                                           \
  static G G##_dbl(G p) {                                                                                            \
    if (ISZ(p.z)) return p;                                                                                          \
    F A = SQR(p.x), B = SQR(p.y), C = SQR(B), t = SQR(ADD(p.x, B));                                                  \
    F D = DBL(SUB(SUB(t, A), C)), E = ADD(DBL(A), A), Fq = SQR(E);                                                   \
    G r; r.x = SUB(Fq, DBL(D)); r.z = DBL(MUL(p.y, p.z));                                                            \
    r.y = SUB(MUL(E, SUB(D, r.x)), DBL(DBL(DBL(C))));                                                                \
    return r;                                                                                                        \
  }                                                                                                                  \
  static G G##_add(G p, G q) {                                                                                       \
    if (ISZ(p.z)) return q;                                                                                          \
    if (ISZ(q.z)) return p;                                                                                          \
    F z1z1 = SQR(p.z), z2z2 = SQR(q.z), u1 = MUL(p.x, z2z2), u2 = MUL(q.x, z1z1);                                    \
    F s1 = MUL(MUL(p.y, q.z), z2z2), s2 = MUL(MUL(q.y, p.z), z1z1);                                                  \
    if (EQ(u1, u2)) { if (EQ(s1, s2)) return G##_dbl(p); G inf = p; memset(&inf.z, 0, sizeof inf.z); return inf; }   \
    F h = SUB(u2, u1), hh = SQR(h), hhh = MUL(h, hh), rr = SUB(s2, s1), v = MUL(u1, hh);                              \
    G r; r.x = SUB(SUB(SQR(rr), hhh), DBL(v));                                                                       \
    r.y = SUB(MUL(rr, SUB(v, r.x)), MUL(s1, hhh));                                                                   \
    r.z = MUL(MUL(p.z, q.z), h);                                                                                     \
    return r;                                                                                                        \
  }                                                                                                                  \
  static G G##_neg(G p) { p.y = NEG(p.y); return p; }                                                                \
  static G G##_mul(G p, const uint64_t* k, int bits) {                                                               \
    G acc = p; memset(&acc.z, 0, sizeof acc.z);                                                                      \
    for (int i = bits - 1; i >= 0; i--) { acc = G##_dbl(acc); if ((k[i >> 6] >> (i & 63)) & 1) acc = G##_add(acc, p); } \
    return acc;                                                                                                      \
  }
static int fpz(fp a) { return fp_is_zero(&a); }
static int fpe(fp a, fp b) { return fp_eq(&a, &b); }
static fp fp_dbl(fp a) { return fp_add(a, a); }
CURVE_TEMPLATE(g1p, fp, fp_add, fp_sub, fp_mul, fp_sqr, fp_dbl, fp_neg, fpz, fpe)
CURVE_TEMPLATE(g2p, fp2, f2_add, f2_sub, f2_mul, f2_sqr, f2_dbl, f2_neg, f2_is_zero, f2_eq)

static void g1_affine(fp* x, fp* y, g1p p) { fp zi = fp_inv(p.z), z2 = fp_sqr(zi); *x = fp_mul(p.x, z2); *y = fp_mul(p.y, fp_mul(z2, zi)); }
static void g2_affine(fp2* x, fp2* y, g2p p) { fp2 zi = f2_inv(p.z), z2 = f2_sqr(zi); *x = f2_mul(p.x, z2); *y = f2_mul(p.y, f2_mul(z2, zi)); }
static void g1_compress(uint8_t* out, g1p p) {
  if (g1p_is_inf(&p)) { memset(out, 0, 48); out[0] = 0xc0; return; }
  fp x, y; g1_affine(&x, &y, p); fp_to_be48(out, x);
  out[0] |= 0x80 | (fp_lex_largest(y) ? 0x20 : 0);
}
static void g2_compress(uint8_t* out, g2p p) {
  if (g2p_is_inf(&p)) { memset(out, 0, 96); out[0] = 0xc0; return; }
  fp2 x, y; g2_affine(&x, &y, p); fp_to_be48(out, x.c1); fp_to_be48(out + 48, x.c0);
  out[0] |= 0x80 | (f2_lex_largest(y) ? 0x20 : 0);
}
static const uint64_t X_ABS[1] = {0xd201000000010000ull};
static fp2 PSI_CX, PSI_CY;
static g2p g2_psi(g2p p) { g2p r = {f2_mul(f2_conj(p.x), PSI_CX), f2_mul(f2_conj(p.y), PSI_CY), f2_conj(p.z)}; return r; }
static g2p g2_mul_x(g2p p) { return g2p_neg(g2p_mul(p, X_ABS, 64)); }       /* [x]P, x negative */
static g2p g2_clear_cofactor(g2p p) {    /* RFC 9380 G.3 */
  g2p t1 = g2_mul_x(p), t2 = g2_psi(p), t3 = g2_psi(g2_psi(g2p_dbl(p)));
  t3 = g2p_add(t3, g2p_neg(t2));
  t2 = g2_mul_x(g2p_add(t1, t2));
  t3 = g2p_add(t3, t2);
  t3 = g2p_add(t3, g2p_neg(t1));
  return g2p_add(t3, g2p_neg(p));
}

/* ------------------------------------------------------------------ SHA-256 */
typedef struct { uint32_t h[8]; uint8_t buf[64]; size_t fill; uint64_t total; } sha256;
static const uint32_t SK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(sha256* s, const uint8_t* p) {
  uint32_t w[64], a = s->h[0], b = s->h[1], c = s->h[2], d = s->h[3], e = s->h[4], f = s->h[5], g = s->h[6], h = s->h[7];
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; i++) w[i] = w[i - 16] + (ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10));
  for (int i = 0; i < 64; i++) {
    uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + SK[i] + w[i];
    uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s->h[0] += a; s->h[1] += b; s->h[2] += c; s->h[3] += d; s->h[4] += e; s->h[5] += f; s->h[6] += g; s->h[7] += h;
}
static void sha_init(sha256* s) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s->h, iv, 32); s->fill = 0; s->total = 0;
}
static void sha_update(sha256* s, const uint8_t* p, size_t n) {
  s->total += n;
  while (n) {
    size_t k = 64 - s->fill < n ? 64 - s->fill : n;
    memcpy(s->buf + s->fill, p, k); s->fill += k; p += k; n -= k;
    if (s->fill == 64) { sha_block(s, s->buf); s->fill = 0; }
  }
}
static void sha_final(sha256* s, uint8_t* out) {
  uint64_t bits = s->total * 8; uint8_t pad[72] = {0x80}, len[8];
  size_t padlen = s->fill < 56 ? 56 - s->fill : 120 - s->fill;
  for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha_update(s, pad, padlen); sha_update(s, len, 8);
  for (int i = 0; i < 8; i++) { out[4 * i] = s->h[i] >> 24; out[4 * i + 1] = s->h[i] >> 16; out[4 * i + 2] = s->h[i] >> 8; out[4 * i + 3] = s->h[i]; }
}

/* ------------------------------------------------------------------ RFC 9380 hash_to_curve */
static void expand_xmd(uint8_t* out, int nout, const uint8_t* pre, size_t prel, const uint8_t* m, size_t ml, const uint8_t* dst, size_t dl) {
  sha256 s; uint8_t b0[32], bi[32] = {0}, z[64] = {0}, t[3] = {(uint8_t)(nout >> 8), (uint8_t)nout, 0}, big[32];
  if (dl > 255) {   /* RFC 9380 5.3.3: DST' = H("H2C-OVERSIZE-DST-" || DST) */
    sha_init(&s); sha_update(&s, (const uint8_t*)"H2C-OVERSIZE-DST-", 17); sha_update(&s, dst, dl); sha_final(&s, big);
    dst = big; dl = 32;
  }
  uint8_t dlb = (uint8_t)dl;
  sha_init(&s); sha_update(&s, z, 64); sha_update(&s, pre, prel); sha_update(&s, m, ml); sha_update(&s, t, 3);
  sha_update(&s, dst, dl); sha_update(&s, &dlb, 1); sha_final(&s, b0);
  for (int blk = 1; blk <= nout / 32; blk++) {
    uint8_t x[33];
    for (int i = 0; i < 32; i++) x[i] = b0[i] ^ bi[i];
    x[32] = (uint8_t)blk;
    sha_init(&s); sha_update(&s, x, 33); sha_update(&s, dst, dl); sha_update(&s, &dlb, 1); sha_final(&s, bi);
    memcpy(out + 32 * (blk - 1), bi, 32);
  }
}
static fp fp_from_be64(const uint8_t* b) {   /* 64-byte big-endian integer mod p: hi * 2^256 + lo */
  fp hi = fp_from_be(b, 32), lo = fp_from_be(b + 32, 32);
  uint64_t w[6] = {0, 0, 0, 0, 1, 0};        /* 2^256 */
  return fp_add(fp_mul(hi, fp_from_plain(w)), lo);
}
static fp ISO1_A, ISO1_B, ISO1_Z;
static fp2 ISO2_A, ISO2_B, ISO2_Z;
/* simplified SWU, RFC 9380 6.6.2 (the non-optimised form) */
static void sswu1(fp* x, fp* y, fp u) {
  fp zu2 = fp_mul(ISO1_Z, fp_sqr(u)), tv1 = fp_add(fp_sqr(zu2), zu2), x1;
  if (fp_is_zero(&tv1)) x1 = fp_mul(ISO1_B, fp_inv(fp_mul(ISO1_Z, ISO1_A)));
  else x1 = fp_mul(fp_mul(fp_neg(ISO1_B), fp_inv(ISO1_A)), fp_add(FP_ONE_M, fp_inv(tv1)));
  fp gx = fp_add(fp_mul(fp_add(fp_sqr(x1), ISO1_A), x1), ISO1_B);
  if (fp_is_square(gx)) { *x = x1; fp_sqrt(y, gx); }
  else { *x = fp_mul(zu2, x1); gx = fp_add(fp_mul(fp_add(fp_sqr(*x), ISO1_A), *x), ISO1_B); fp_sqrt(y, gx); }
  if (fp_parity(u) != fp_parity(*y)) *y = fp_neg(*y);
}
static void sswu2(fp2* x, fp2* y, fp2 u) {
  fp2 zu2 = f2_mul(ISO2_Z, f2_sqr(u)), tv1 = f2_add(f2_sqr(zu2), zu2), x1;
  if (f2_is_zero(tv1)) x1 = f2_mul(ISO2_B, f2_inv(f2_mul(ISO2_Z, ISO2_A)));
  else x1 = f2_mul(f2_mul(f2_neg(ISO2_B), f2_inv(ISO2_A)), f2_add(FP2_ONE, f2_inv(tv1)));
  fp2 gx = f2_add(f2_mul(f2_add(f2_sqr(x1), ISO2_A), x1), ISO2_B);
  if (f2_is_square(gx)) { *x = x1; f2_sqrt(y, gx); }
  else { *x = f2_mul(zu2, x1); gx = f2_add(f2_mul(f2_add(f2_sqr(*x), ISO2_A), *x), ISO2_B); f2_sqrt(y, gx); }
  if (f2_sgn0(u) != f2_sgn0(*y)) *y = f2_neg(*y);
}
static fp horner1(const uint64_t (*k)[6], int n, fp x) { fp acc = fp_from_plain(k[n - 1]); for (int i = n - 2; i >= 0; i--) acc = fp_add(fp_mul(acc, x), fp_from_plain(k[i])); return acc; }
static fp2 horner2(const uint64_t (*k)[12], int n, fp2 x) { fp2 acc = f2_from_plain(k[n - 1]); for (int i = n - 2; i >= 0; i--) acc = f2_add(f2_mul(acc, x), f2_from_plain(k[i])); return acc; }
static g1p iso1(fp x, fp y) {
  fp xd = horner1(K_ISO1_XD, 11, x), yd = horner1(K_ISO1_YD, 16, x);
  g1p r; memset(&r, 0, sizeof r);
  if (fp_is_zero(&xd) || fp_is_zero(&yd)) { r.x = FP_ONE_M; r.y = FP_ONE_M; return r; }
  r.x = fp_mul(horner1(K_ISO1_XN, 12, x), fp_inv(xd));
  r.y = fp_mul(y, fp_mul(horner1(K_ISO1_YN, 16, x), fp_inv(yd)));
  r.z = FP_ONE_M;
  return r;
}
static g2p iso2(fp2 x, fp2 y) {
  fp2 xd = horner2(K_ISO2_XD, 3, x), yd = horner2(K_ISO2_YD, 4, x);
  g2p r; memset(&r, 0, sizeof r);
  if (f2_is_zero(xd) || f2_is_zero(yd)) { r.x = FP2_ONE; r.y = FP2_ONE; return r; }
  r.x = f2_mul(horner2(K_ISO2_XN, 4, x), f2_inv(xd));
  r.y = f2_mul(y, f2_mul(horner2(K_ISO2_YN, 4, x), f2_inv(yd)));
  r.z = FP2_ONE;
  return r;
}
static g1p hash_to_g1(const uint8_t* pre, size_t prel, const uint8_t* m, size_t ml, const uint8_t* dst, size_t dl) {
  uint8_t ub[128]; fp x, y;
  expand_xmd(ub, 128, pre, prel, m, ml, dst, dl);
  sswu1(&x, &y, fp_from_be64(ub)); g1p q0 = iso1(x, y);
  sswu1(&x, &y, fp_from_be64(ub + 64)); g1p q1 = iso1(x, y);
  g1p q = g1p_add(q0, q1);
  return g1p_add(g1p_mul(q, X_ABS, 64), q);           /* h_eff = 1 - x = 1 + |x| */
}
static g2p hash_to_g2(const uint8_t* pre, size_t prel, const uint8_t* m, size_t ml, const uint8_t* dst, size_t dl) {
  uint8_t ub[256]; fp2 x, y, u;
  expand_xmd(ub, 256, pre, prel, m, ml, dst, dl);
  u.c0 = fp_from_be64(ub); u.c1 = fp_from_be64(ub + 64); sswu2(&x, &y, u); g2p q0 = iso2(x, y);
  u.c0 = fp_from_be64(ub + 128); u.c1 = fp_from_be64(ub + 192); sswu2(&x, &y, u); g2p q1 = iso2(x, y);
  return g2_clear_cofactor(g2p_add(q0, q1));
}

/* ------------------------------------------------------------------ pairing: Jacobian twist coordinates
 * tangent at T=(X,Y,Z), scaled by 2YZ^3:  l0 = 3X^3 - 2Y^2,  l2 = -3X^2 Z^2 xP,  l3 = 2YZ^3 yP
 * chord T,Q=(x2,y2), scaled by Z3 = Z H:   l0 = r x2 - y2 Z3,  l2 = -r xP,        l3 = Z3 yP      (r = y2 Z^3 - Y, H = x2 Z^2 - X) */
static fp12 miller_loop(int n, const fp* xp, const fp* yp, const fp2* xq, const fp2* yq) {
  g2p T8[8]; g2p* T = n <= 8 ? T8 : (g2p*)xmalloc(sizeof(g2p) * (size_t)n); fp12 f = F12_ONE;
  for (int k = 0; k < n; k++) { T[k].x = xq[k]; T[k].y = yq[k]; T[k].z = FP2_ONE; }
  for (int i = 62; i >= 0; i--) {
    f = f12_sqr(f);
    for (int k = 0; k < n; k++) {
      fp2 X = T[k].x, Y = T[k].y, Z = T[k].z, A = f2_sqr(X), B = f2_sqr(Y), E = f2_add(f2_dbl(A), A), ZZ = f2_sqr(Z);
      T[k] = g2p_dbl(T[k]);
      fp2 l0 = f2_sub(f2_mul(E, X), f2_dbl(B)), l2 = f2_neg(f2_mul_fp(f2_mul(E, ZZ), xp[k])), l3 = f2_mul_fp(f2_mul(T[k].z, ZZ), yp[k]);
      f = f12_mul_line(f, l0, l2, l3);
    }
    if ((X_ABS[0] >> i) & 1) {
      for (int k = 0; k < n; k++) {
        fp2 Z = T[k].z, ZZ = f2_sqr(Z), H = f2_sub(f2_mul(xq[k], ZZ), T[k].x), r = f2_sub(f2_mul(yq[k], f2_mul(Z, ZZ)), T[k].y);
        fp2 HH = f2_sqr(H), HHH = f2_mul(H, HH), V = f2_mul(T[k].x, HH), Z3 = f2_mul(Z, H);
        g2p n3; n3.x = f2_sub(f2_sub(f2_sqr(r), HHH), f2_dbl(V)); n3.y = f2_sub(f2_mul(r, f2_sub(V, n3.x)), f2_mul(T[k].y, HHH)); n3.z = Z3;
        fp2 l0 = f2_sub(f2_mul(r, xq[k]), f2_mul(yq[k], Z3)), l2 = f2_neg(f2_mul_fp(r, xp[k])), l3 = f2_mul_fp(Z3, yp[k]);
        T[k] = n3;
        f = f12_mul_line(f, l0, l2, l3);
      }
    }
  }
  if (T != T8) free(T);
  return f12_conj(f);
}
static fp12 pow_x(fp12 a) {
  fp12 acc = a;
  for (int i = 62; i >= 0; i--) { acc = f12_cyc_sqr(acc); if ((X_ABS[0] >> i) & 1) acc = f12_mul(acc, a); }
  return f12_conj(acc);
}
static fp12 final_exp(fp12 fin) {        /* cube of the canonical value: 3(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3 */
  fp12 f = f12_mul(f12_conj(fin), f12_inv(fin));
  f = f12_mul(f12_frob(f, 2), f);
  fp12 t = f12_mul(pow_x(f), f12_conj(f));
  t = f12_mul(pow_x(t), f12_conj(t));
  t = f12_mul(pow_x(t), f12_frob(t, 1));
  t = f12_mul(f12_mul(pow_x(pow_x(t)), f12_frob(t, 2)), f12_conj(t));
  return f12_mul(t, f12_mul(f12_cyc_sqr(f), f));
}

/* ------------------------------------------------------------------ reference control flow */
static const char* DSTS[2][3] = {
    {"BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_", "BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_AUG_", "BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_"},
    {"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_", "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_AUG_", "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"}};
static g1p G1_GEN; static g2p G2_GEN;
static g1p load_g1(const uint8_t* raw) { g1p p; memcpy(&p, raw, 144); return p; }      /* RAW_PROJ = Montgomery limbs */
static g2p load_g2(const uint8_t* raw) { g2p p; memcpy(&p, raw, 288); return p; }

/* core_verify, reference src/traits/sig_core.rs:120-146 (aug_prefix: src/traits/sig_aug.rs:20-24) */
static int core_verify(int sg, int scheme, int aug_prefix, const uint8_t* pk_raw, const uint8_t* sig_raw, const uint8_t* msg, size_t len) {
  const uint8_t* dst = (const uint8_t*)DSTS[sg - 1][scheme]; size_t dl = strlen((const char*)dst);
  fp xp[2], yp[2]; fp2 xq[2], yq[2]; uint8_t pre[96]; size_t prel = 0;
  if (sg == 1) {
    g2p pk = load_g2(pk_raw); g1p sig = load_g1(sig_raw);
    if (g1p_is_inf(&sig)) return 2;
    if (g2p_is_inf(&pk)) return 3;
    if (aug_prefix) { g2_compress(pre, pk); prel = 96; }
    g1p h = hash_to_g1(pre, prel, msg, len, dst, dl);
    g1_affine(&xp[0], &yp[0], h); g2_affine(&xq[0], &yq[0], pk);
    g1_affine(&xp[1], &yp[1], sig); g2_affine(&xq[1], &yq[1], g2p_neg(G2_GEN));
  } else {
    g1p pk = load_g1(pk_raw); g2p sig = load_g2(sig_raw);
    if (g2p_is_inf(&sig)) return 2;
    if (g1p_is_inf(&pk)) return 3;
    if (aug_prefix) { g1_compress(pre, pk); prel = 48; }
    g2p h = hash_to_g2(pre, prel, msg, len, dst, dl);
    g1_affine(&xp[0], &yp[0], pk); g2_affine(&xq[0], &yq[0], h);            /* G1 member first: src/helpers.rs:53-63 */
    g1_affine(&xp[1], &yp[1], g1p_neg(G1_GEN)); g2_affine(&xq[1], &yq[1], sig);
  }
  return f12_is_one(final_exp(miller_loop(2, xp, yp, xq, yq))) ? 0 : 1;
}

void bo_init(void) {
  if (g_init) return;
  memset(&FP_ZERO, 0, sizeof FP_ZERO);
  memcpy(FP_R2.l, K_R2, 48);
  uint64_t one[6] = {1, 0, 0, 0, 0, 0};
  FP_ONE_M = fp_from_plain(one);
  FP2_ZERO.c0 = FP_ZERO; FP2_ZERO.c1 = FP_ZERO; FP2_ONE.c0 = FP_ONE_M; FP2_ONE.c1 = FP_ZERO;
  memset(&F12_ONE, 0, sizeof F12_ONE); F12_ONE.c0.a0 = FP2_ONE;
  for (int k = 0; k < 6; k++) { FROB[0][k] = f2_from_plain(K_FROB1[k]); FROB[1][k] = f2_from_plain(K_FROB2[k]); }
  PSI_CX = f2_from_plain(K_PSI_CX); PSI_CY = f2_from_plain(K_PSI_CY);
  ISO1_A = fp_from_plain(K_ISO1_A); ISO1_B = fp_from_plain(K_ISO1_B); ISO1_Z = fp_from_plain(K_ISO1_Z);
  ISO2_A = f2_from_plain(K_ISO2_A); ISO2_B = f2_from_plain(K_ISO2_B); ISO2_Z = f2_from_plain(K_ISO2_Z);
  G1_GEN.x = fp_from_plain(K_G1X); G1_GEN.y = fp_from_plain(K_G1Y); G1_GEN.z = FP_ONE_M;
  G2_GEN.x = f2_from_plain(K_G2X); G2_GEN.y = f2_from_plain(K_G2Y); G2_GEN.z = FP2_ONE;
  g_init = 1;
}

/* Signature::verify, reference src/signature.rs:130-138 */
int bo_verify(int sg, int scheme, const uint8_t* pk, const uint8_t* sig, const uint8_t* msg, size_t len) {
  bo_init();
  return core_verify(sg, scheme, scheme == 1, pk, sig, msg, len);
}

typedef struct { int sg, scheme; const uint8_t *pks, *sigs, *msgs; const uint64_t* offs; size_t lo, hi; int32_t* st; } job;
static void* worker(void* a) {
  job* j = (job*)a;
  size_t pks = j->sg == 1 ? 288 : 144, sgs = j->sg == 1 ? 144 : 288;
  for (size_t i = j->lo; i < j->hi; i++)
    j->st[i] = core_verify(j->sg, j->scheme, j->scheme == 1, j->pks + i * pks, j->sigs + i * sgs, j->msgs + j->offs[i], (size_t)(j->offs[i + 1] - j->offs[i]));
  return 0;
}
/* n independent Signature::verify calls on `threads` host threads (contiguous item ranges) */
void bo_verify_batch(int sg, int scheme, const uint8_t* pks, const uint8_t* sigs, const uint8_t* msgs, const uint64_t* offs, size_t n, int32_t* st, int threads) {
  bo_init();
  if (threads < 1) threads = 1;
  bo_thread* th = (bo_thread*)xmalloc(sizeof(bo_thread) * threads);
  job* jobs = (job*)xmalloc(sizeof(job) * threads);
  for (int t = 0; t < threads; t++) {
    job j = {sg, scheme, pks, sigs, msgs, offs, n * t / threads, n * (t + 1) / threads, st};
    jobs[t] = j;
    bo_spawn(&th[t], worker, &jobs[t], threads == 1);
  }
  for (int t = 0; t < threads; t++) bo_join(&th[t]);
  free(th); free(jobs);
}

void bo_hash_to_point(int group, const uint8_t* m, size_t ml, const uint8_t* dst, size_t dl, uint8_t* out) {
  bo_init();
  if (group == 1) g1_compress(out, hash_to_g1(0, 0, m, ml, dst, dl)); else g2_compress(out, hash_to_g2(0, 0, m, ml, dst, dl));
}
void bo_compress(int group, const uint8_t* raw, uint8_t* out) { bo_init(); if (group == 1) g1_compress(out, load_g1(raw)); else g2_compress(out, load_g2(raw)); }

/* verify_secure, reference src/secure_aggregation.rs:173-208 with hash_public_keys_with_sorted (:37-106).
 * legacy != 0: the Dash header transcode of src/impls/legacy.rs:19-35 on 48-byte keys (sg == 2 only). */
static size_t g_width;
static int cmp_keys(const void* a, const void* b) {
  const uint8_t* x = *(const uint8_t* const*)a; const uint8_t* y = *(const uint8_t* const*)b;
  int c = memcmp(x, y, g_width);
  return c ? c : (x < y ? -1 : x > y);     /* stable: ties keep input order (keys sit in one array) */
}
int bo_verify_secure(int sg, int scheme, const uint8_t* pks, size_t n, const uint8_t* sig_raw, const uint8_t* msg, size_t len, int legacy) {
  bo_init();
  size_t psz = sg == 1 ? 288 : 144, width = sg == 1 ? 96 : 48;
  if (n == 0) {
    int inf;
    if (sg == 1) { g1p s1 = load_g1(sig_raw); inf = g1p_is_inf(&s1); } else { g2p s2 = load_g2(sig_raw); inf = g2p_is_inf(&s2); }
    return inf ? 0 : 1;
  }
  uint8_t* kb = (uint8_t*)xmalloc(width * n); const uint8_t** order = (const uint8_t**)xmalloc(sizeof(void*) * n);
  for (size_t i = 0; i < n; i++) {
    if (sg == 1) g2_compress(kb + i * width, load_g2(pks + i * psz)); else g1_compress(kb + i * width, load_g1(pks + i * psz));
    if (legacy && kb[i * width] != 0xc0) { uint8_t ys = kb[i * width] & 0x20; kb[i * width] &= 0x1f; if (ys) kb[i * width] |= 0x80; }
    order[i] = kb + i * width;
  }
  g_width = width; qsort(order, n, sizeof(void*), cmp_keys);
  sha256 s; uint8_t H[32]; sha_init(&s);
  for (size_t i = 0; i < n; i++) sha_update(&s, order[i], width);
  sha_final(&s, H);
  g1p a1; g2p a2; memset(&a1, 0, sizeof a1); memset(&a2, 0, sizeof a2);
  int rc = -1;
  for (size_t i = 0; i < n; i++) {
    uint8_t buf[36] = {(uint8_t)(i >> 24), (uint8_t)(i >> 16), (uint8_t)(i >> 8), (uint8_t)i}, d[32];
    memcpy(buf + 4, H, 32); sha_init(&s); sha_update(&s, buf, 36); sha_final(&s, d);
    uint64_t t[4]; for (int k = 0; k < 4; k++) { t[k] = 0; for (int b = 0; b < 8; b++) t[k] |= (uint64_t)d[31 - 8 * k - b] << (8 * b); }
    for (int round = 0; round < 3; round++) {                      /* int_BE(hash) mod r */
      uint64_t dd[4]; u128 bw = 0;
      for (int k = 0; k < 4; k++) { u128 x = (u128)t[k] - K_R[k] - bw; dd[k] = (uint64_t)x; bw = (x >> 64) & 1; }
      if (bw) break;
      memcpy(t, dd, 32);
    }
    if (!(t[0] | t[1] | t[2] | t[3])) { rc = 5; break; }
    size_t idx = (size_t)(order[i] - kb) / width;
    if (sg == 1) a2 = g2p_add(a2, g2p_mul(load_g2(pks + idx * psz), t, 256)); else a1 = g1p_add(a1, g1p_mul(load_g1(pks + idx * psz), t, 256));
  }
  free(kb); free(order);
  if (rc >= 0) return rc;
  return core_verify(sg, scheme, 0, sg == 1 ? (const uint8_t*)&a2 : (const uint8_t*)&a1, sig_raw, msg, len);
}

/* ------------------------------------------------------------------ CPU legs of BASELINE configs 3, 4, 5 (tools/bench_configs.py)
 * threads == 1 is the reference's own loop; threads > 1 splits the items into contiguous ranges whose partial results
 * (point sums, Miller products) are folded at the end -- the fairest multi-core form of the same algorithm. */
typedef struct { int sg; const uint8_t* pks; size_t lo, hi; g1p s1; g2p s2; } sum_job;
static void* sum_worker(void* a) {
  sum_job* j = (sum_job*)a;
  memset(&j->s1, 0, sizeof j->s1); memset(&j->s2, 0, sizeof j->s2);
  for (size_t i = j->lo; i < j->hi; i++) {               /* reference src/traits/pk_multi.rs:7-13: g += key */
    if (j->sg == 1) j->s2 = g2p_add(j->s2, load_g2(j->pks + i * 288)); else j->s1 = g1p_add(j->s1, load_g1(j->pks + i * 144));
  }
  return 0;
}
/* MultiSignature::verify (reference src/multi_signature.rs:127-135) with MultiPublicKey::from_public_keys */
int bo_multi_verify(int sg, int scheme, const uint8_t* pks, size_t n, const uint8_t* sig, const uint8_t* msg, size_t len, int threads) {
  bo_init();
  if (threads < 1) threads = 1;
  bo_thread* th = (bo_thread*)xmalloc(sizeof(bo_thread) * threads);
  sum_job* jobs = (sum_job*)xmalloc(sizeof(sum_job) * threads);
  for (int t = 0; t < threads; t++) {
    jobs[t].sg = sg; jobs[t].pks = pks; jobs[t].lo = n * t / threads; jobs[t].hi = n * (t + 1) / threads;
    bo_spawn(&th[t], sum_worker, &jobs[t], threads == 1);
  }
  g1p a1; g2p a2; memset(&a1, 0, sizeof a1); memset(&a2, 0, sizeof a2);
  for (int t = 0; t < threads; t++) {
    bo_join(&th[t]);
    a1 = g1p_add(a1, jobs[t].s1); a2 = g2p_add(a2, jobs[t].s2);
  }
  free(th); free(jobs);
  return core_verify(sg, scheme, scheme == 1, sg == 1 ? (const uint8_t*)&a2 : (const uint8_t*)&a1, sig, msg, len);
}

typedef struct { int sg, scheme; const uint8_t *pks, *msgs; const uint64_t* offs; size_t lo, hi; fp12 f; long first_bad; } agg_job;
static void* agg_worker(void* a) {
  agg_job* j = (agg_job*)a;
  const uint8_t* dst = (const uint8_t*)DSTS[j->sg - 1][j->scheme]; size_t dl = strlen((const char*)dst);
  size_t k = j->hi - j->lo;
  fp* xp = (fp*)xmalloc(sizeof(fp) * (k + 1)); fp* yp = (fp*)xmalloc(sizeof(fp) * (k + 1));
  fp2* xq = (fp2*)xmalloc(sizeof(fp2) * (k + 1)); fp2* yq = (fp2*)xmalloc(sizeof(fp2) * (k + 1));
  j->first_bad = -1;
  size_t m = 0;
  for (size_t i = j->lo; i < j->hi; i++) {               /* reference src/traits/sig_core.rs:161-171 */
    uint8_t pre[96]; size_t prel = 0;
    const uint8_t* msg = j->msgs + j->offs[i]; size_t len = (size_t)(j->offs[i + 1] - j->offs[i]);
    if (j->sg == 1) {
      g2p pk = load_g2(j->pks + i * 288);
      if (g2p_is_inf(&pk)) { if (j->first_bad < 0) j->first_bad = (long)i; continue; }
      if (j->scheme == 1) { g2_compress(pre, pk); prel = 96; }
      g1_affine(&xp[m], &yp[m], hash_to_g1(pre, prel, msg, len, dst, dl)); g2_affine(&xq[m], &yq[m], pk);
    } else {
      g1p pk = load_g1(j->pks + i * 144);
      if (g1p_is_inf(&pk)) { if (j->first_bad < 0) j->first_bad = (long)i; continue; }
      if (j->scheme == 1) { g1_compress(pre, pk); prel = 48; }
      g1_affine(&xp[m], &yp[m], pk); g2_affine(&xq[m], &yq[m], hash_to_g2(pre, prel, msg, len, dst, dl));
    }
    m++;
  }
  j->f = m ? miller_loop((int)m, xp, yp, xq, yq) : F12_ONE;
  free(xp); free(yp); free(xq); free(yq);
  return 0;
}
static const uint8_t* g_dup_msgs; static const uint64_t* g_dup_offs;
static int cmp_msgs(const void* a, const void* b) {
  size_t i = *(const size_t*)a, k = *(const size_t*)b;
  size_t li = (size_t)(g_dup_offs[i + 1] - g_dup_offs[i]), lk = (size_t)(g_dup_offs[k + 1] - g_dup_offs[k]);
  int c = memcmp(g_dup_msgs + g_dup_offs[i], g_dup_msgs + g_dup_offs[k], li < lk ? li : lk);
  if (c) return c;
  if (li != lk) return li < lk ? -1 : 1;
  return i < k ? -1 : i > k;
}
/* AggregateSignature::verify (reference src/aggregate_signature.rs:230-239): Basic's duplicate rule (src/traits/sig_basic.rs:46-58),
 * then core_aggregate_verify (src/traits/sig_core.rs:149-178).  Returns the status code; aux = the indices of the error strings. */
int bo_aggregate_verify(int sg, int scheme, const uint8_t* pks, const uint8_t* msgs, const uint64_t* offs, size_t n, const uint8_t* sig_raw,
                        int threads, uint64_t* aux) {
  bo_init();
  aux[0] = aux[1] = 0;
  if (scheme == 0 && n > 1) {      /* first i whose message equals an earlier one: sort indices by (message, index), scan runs */
    size_t* idx = (size_t*)xmalloc(sizeof(size_t) * n);
    for (size_t i = 0; i < n; i++) idx[i] = i;
    g_dup_msgs = msgs; g_dup_offs = offs; qsort(idx, n, sizeof(size_t), cmp_msgs);
    size_t best = (size_t)-1, old = 0, run = 0;
    for (size_t k = 1; k < n; k++) {
      size_t a = idx[k - 1], b = idx[k];
      size_t la = (size_t)(offs[a + 1] - offs[a]), lb = (size_t)(offs[b + 1] - offs[b]);
      if (la == lb && memcmp(msgs + offs[a], msgs + offs[b], la) == 0) { if (b < best) { best = b; old = idx[run]; } }
      else run = k;
    }
    free(idx);
    if (best != (size_t)-1) { aux[0] = old; aux[1] = best; return 4; }
  }
  if (sg == 1) { g1p s = load_g1(sig_raw); if (g1p_is_inf(&s)) return 2; } else { g2p s = load_g2(sig_raw); if (g2p_is_inf(&s)) return 2; }
  if (threads < 1) threads = 1;
  bo_thread* th = (bo_thread*)xmalloc(sizeof(bo_thread) * threads);
  agg_job* jobs = (agg_job*)xmalloc(sizeof(agg_job) * threads);
  for (int t = 0; t < threads; t++) {
    agg_job j = {sg, scheme, pks, msgs, offs, n * t / threads, n * (t + 1) / threads, F12_ONE, -1};
    jobs[t] = j;
    bo_spawn(&th[t], agg_worker, &jobs[t], threads == 1);
  }
  fp12 f = F12_ONE; long fb = -1;
  for (int t = 0; t < threads; t++) {
    bo_join(&th[t]);
    if (fb < 0 && jobs[t].first_bad >= 0) fb = jobs[t].first_bad;
    f = f12_mul(f, jobs[t].f);
  }
  free(th); free(jobs);
  if (fb >= 0) { aux[0] = (uint64_t)fb + 1; return 3; }
  fp xp, yp; fp2 xq, yq;
  if (sg == 1) { g1_affine(&xp, &yp, load_g1(sig_raw)); g2_affine(&xq, &yq, g2p_neg(G2_GEN)); }
  else { g1_affine(&xp, &yp, g1p_neg(G1_GEN)); g2_affine(&xq, &yq, load_g2(sig_raw)); }
  f = f12_mul(f, miller_loop(1, &xp, &yp, &xq, &yq));
  return f12_is_one(final_exp(f)) ? 0 : 1;
}

typedef struct { int sg; const uint8_t* pks; const uint8_t* const* order; const uint8_t* kb; size_t width; const uint8_t* H; size_t lo, hi; g1p s1; g2p s2; int zero; } sec_job;
static void* sec_worker(void* a) {
  sec_job* j = (sec_job*)a;
  size_t psz = j->sg == 1 ? 288 : 144;
  memset(&j->s1, 0, sizeof j->s1); memset(&j->s2, 0, sizeof j->s2); j->zero = 0;
  for (size_t i = j->lo; i < j->hi; i++) {               /* reference src/secure_aggregation.rs:61-100,201-204 */
    sha256 s; uint8_t buf[36] = {(uint8_t)(i >> 24), (uint8_t)(i >> 16), (uint8_t)(i >> 8), (uint8_t)i}, d[32];
    memcpy(buf + 4, j->H, 32); sha_init(&s); sha_update(&s, buf, 36); sha_final(&s, d);
    uint64_t t[4]; for (int k = 0; k < 4; k++) { t[k] = 0; for (int b = 0; b < 8; b++) t[k] |= (uint64_t)d[31 - 8 * k - b] << (8 * b); }
    for (int round = 0; round < 3; round++) {
      uint64_t dd[4]; u128 bw = 0;
      for (int k = 0; k < 4; k++) { u128 x = (u128)t[k] - K_R[k] - bw; dd[k] = (uint64_t)x; bw = (x >> 64) & 1; }
      if (bw) break;
      memcpy(t, dd, 32);
    }
    if (!(t[0] | t[1] | t[2] | t[3])) { j->zero = 1; return 0; }
    size_t idx = (size_t)(j->order[i] - j->kb) / j->width;
    if (j->sg == 1) j->s2 = g2p_add(j->s2, g2p_mul(load_g2(j->pks + idx * psz), t, 256)); else j->s1 = g1p_add(j->s1, g1p_mul(load_g1(j->pks + idx * psz), t, 256));
  }
  return 0;
}
/* verify_secure as bo_verify_secure, the n scalar multiplications spread over `threads` host threads */
int bo_verify_secure_mt(int sg, int scheme, const uint8_t* pks, size_t n, const uint8_t* sig_raw, const uint8_t* msg, size_t len, int legacy, int threads) {
  bo_init();
  if (n == 0 || threads <= 1) return bo_verify_secure(sg, scheme, pks, n, sig_raw, msg, len, legacy);
  size_t psz = sg == 1 ? 288 : 144, width = sg == 1 ? 96 : 48;
  uint8_t* kb = (uint8_t*)xmalloc(width * n); const uint8_t** order = (const uint8_t**)xmalloc(sizeof(void*) * n);
  for (size_t i = 0; i < n; i++) {
    if (sg == 1) g2_compress(kb + i * width, load_g2(pks + i * psz)); else g1_compress(kb + i * width, load_g1(pks + i * psz));
    if (legacy && kb[i * width] != 0xc0) { uint8_t ys = kb[i * width] & 0x20; kb[i * width] &= 0x1f; if (ys) kb[i * width] |= 0x80; }
    order[i] = kb + i * width;
  }
  g_width = width; qsort(order, n, sizeof(void*), cmp_keys);
  sha256 s; uint8_t H[32]; sha_init(&s);
  for (size_t i = 0; i < n; i++) sha_update(&s, order[i], width);
  sha_final(&s, H);
  bo_thread* th = (bo_thread*)xmalloc(sizeof(bo_thread) * threads);
  sec_job* jobs = (sec_job*)xmalloc(sizeof(sec_job) * threads);
  for (int t = 0; t < threads; t++) {
    sec_job j; memset(&j, 0, sizeof j);
    j.sg = sg; j.pks = pks; j.order = order; j.kb = kb; j.width = width; j.H = H; j.lo = n * t / threads; j.hi = n * (t + 1) / threads;
    jobs[t] = j;
    bo_spawn(&th[t], sec_worker, &jobs[t], 0);
  }
  g1p a1; g2p a2; memset(&a1, 0, sizeof a1); memset(&a2, 0, sizeof a2);
  int zero = 0;
  for (int t = 0; t < threads; t++) {
    bo_join(&th[t]);
    zero |= jobs[t].zero;
    a1 = g1p_add(a1, jobs[t].s1); a2 = g2p_add(a2, jobs[t].s2);
  }
  free(th); free(jobs); free(kb); free(order);
  if (zero) return 5;
  return core_verify(sg, scheme, 0, sg == 1 ? (const uint8_t*)&a2 : (const uint8_t*)&a1, sig_raw, msg, len);
}
