// Fp2 over DPP rows: tower.cuh's fp2 interface on the row-wide field type `wf` (csrc/wide.cuh), so that the templated SSWU map
// and isogeny of hash-to-G2 (h2c.cuh sswu_g2, iso_map_g2_hom) run one map per row.  Same bound contract as the lane-local
// layer: products take sums of at most two normalised values, every result is normalised (one carry pass, five instructions
// beside a 240-instruction product).  What needs the canonical form or is rare (sign, the real-w square root) converts the
// row's element to one lane's lane-local code through LDS.
#pragma once
#include "tower.cuh"
#include "wide.cuh"

__device__ __noinline__ void fp_inv(wf& r, const wf& a) {       // every lane of the row runs the lane-local inversion on the same value
  fp t;
  wf_to_local(t, a);
  fp_inv(t, t);
  fp_reduce(t, t);
  wf_from_local(r, t);
}
struct wf2 {
  wf c0, c1;
};
__device__ __forceinline__ void fp2_load(wf2& r, const uint32_t* c) {
  fp_load(r.c0, c);
  fp_load(r.c1, c + FP_NL);
}
__device__ __forceinline__ void fp2_one(wf2& r) {
  fp_one(r.c0);
  fp_zero(r.c1);
}
__device__ __forceinline__ bool fp2_is_zero(const wf2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
__device__ __forceinline__ void fp2_cmov(wf2& r, const wf2& a, bool c) {
  fp_cmov(r.c0, a.c0, c);
  fp_cmov(r.c1, a.c1, c);
}
__device__ __forceinline__ void fp2_add(wf2& r, const wf2& a, const wf2& b) {
  r.c0.v = wf_norm1(a.c0.v + b.c0.v);
  r.c1.v = wf_norm1(a.c1.v + b.c1.v);
}
__device__ __forceinline__ void fp2_sub(wf2& r, const wf2& a, const wf2& b) {
  r.c0.v = wf_norm1(a.c0.v - b.c0.v);
  r.c1.v = wf_norm1(a.c1.v - b.c1.v);
}
__device__ __forceinline__ void fp2_neg(wf2& r, const wf2& a) {
  fp_neg(r.c0, a.c0);
  fp_neg(r.c1, a.c1);
}
__device__ __forceinline__ void fp2_conj(wf2& r, const wf2& a) {
  r.c0 = a.c0;
  fp_neg(r.c1, a.c1);
}
__device__ __forceinline__ void fp2_reduce(wf2& r, const wf2& a) {
  fp_reduce(r.c0, a.c0);
  fp_reduce(r.c1, a.c1);
}
__device__ __forceinline__ void fp2_mul(wf2& r, const wf2& a, const wf2& b) {      // Karatsuba
  const wfp t0 = wf_mul_leaf(a.c0.v, b.c0.v), t1 = wf_mul_leaf(a.c1.v, b.c1.v);
  const wfp m = wf_mul_leaf(wf_norm1(a.c0.v + a.c1.v), wf_norm1(b.c0.v + b.c1.v));
  r.c0.v = wf_norm1(t0 - t1);
  r.c1.v = wf_norm1(m - t0 - t1);
}
__device__ __forceinline__ void fp2_sqr(wf2& r, const wf2& a) {                    // complex squaring
  const wfp m = wf_mul_leaf(a.c0.v, a.c1.v);
  r.c0.v = wf_mul_leaf(wf_norm1(a.c0.v + a.c1.v), wf_norm1(a.c0.v - a.c1.v));
  r.c1.v = wf_norm1(m + m);
}
__device__ __forceinline__ void fp2_mul_fp(wf2& r, const wf2& a, const wf& k) {
  fp_mul(r.c0, a.c0, k);
  fp_mul(r.c1, a.c1, k);
}
__device__ __noinline__ uint32_t fp2_sgn0(const wf2& a) {
  fp2 t;
  wf_to_local(t.c0, a.c0);
  wf_to_local(t.c1, a.c1);
  return fp2_sgn0(t);
}
// the generic square root (with the shared inversion of *e) on one lane's code: sswu_g2 takes it for a real operand only
__device__ __noinline__ bool fp2_sqrt_inv(wf2& r, const wf2& a, const wf* e, wf* einv) {
  wf2 ar;
  wf er;
  fp2_reduce(ar, a);
  fp_reduce(er, *e);
  fp2 la, lr;
  fp le, li;
  wf_to_local(la.c0, ar.c0);
  wf_to_local(la.c1, ar.c1);
  wf_to_local(le, er);
  const bool ok = fp2_sqrt_inv(lr, la, &le, &li);
  fp_reduce(lr.c0, lr.c0);
  fp_reduce(lr.c1, lr.c1);
  fp_reduce(li, li);
  wf_from_local(r.c0, lr.c0);
  wf_from_local(r.c1, lr.c1);
  wf_from_local(*einv, li);
  return ok;
}
