// __global__ kernels of the verify path.  One lane = one item; intermediates live in HBM workspaces laid out
// word-major (word k of item i at ws[k * stride + i]) so that every workspace access is a coalesced dword stream.
#pragma once
#ifndef BLS_SPLIT_WAVES
#define BLS_SPLIT_WAVES 2   // waves per SIMD the lane-split kernels are compiled for (register budget 512 / waves)
#endif
#include "verify.cuh"

#define BLS_BLOCK 64           // one wave per workgroup: 65,536 items = 1,024 waves = one per SIMD
#define W1 FP_NL               // workspace words per Fp (internal 28-bit-limb form, fp.cuh)
#define W2 (2 * FP_NL)         // per Fp2
#define WS_PAIRS_WORDS (6 * W2)  // P0(2 Fp) Q0(2 Fp2) P1 Q1, affine, internal form
#define WS_PAIR1_WORDS (3 * W2)  // one (P, Q) pair: the one-pair-per-item workspaces of aggregate verify / pairing products
#define WS_F_WORDS (6 * W2)      // Fp12
#define LINE5_WORDS (5 * FP_NL)  // one lane's share of a merged line value (tower.cuh line5_t), see k_lines2s
#define LINE3_WORDS_H (3 * FP_NL) // ... and of one pair's plain line value (two general pairs: first pass of k_lines2s)
#define FX_STORE_WORDS (10 * 6 * FP_NL)   // one lane's value store of k_finalexp2s (VS_SLOTS slots of an Fp12 share)
#define MILLER1_GROUP 3          // items per Miller loop in the pairing-product kernel (k_miller1s): 262,144 pairs take 29.4 / 28.8 ms with 2 / 3; 4 was measured slower (state of four points spills)

struct dst_arg {
  uint8_t b[256];
  uint32_t len;
};

// ---- word-major workspace accessors
__device__ __forceinline__ void ws_ld_fp(fp& r, const uint32_t* ws, size_t stride, size_t i, int w0) {
#pragma unroll
  for (int k = 0; k < FP_NL; k++) r.l[k] = (int32_t)ws[(size_t)(w0 + k) * stride + i];
}
__device__ __forceinline__ void ws_st_fp(uint32_t* ws, size_t stride, size_t i, int w0, const fp& a) {
#pragma unroll
  for (int k = 0; k < FP_NL; k++) ws[(size_t)(w0 + k) * stride + i] = (uint32_t)a.l[k];
}
__device__ __forceinline__ void ws_ld_fp2(fp2& r, const uint32_t* ws, size_t stride, size_t i, int w0) {
  ws_ld_fp(r.c0, ws, stride, i, w0);
  ws_ld_fp(r.c1, ws, stride, i, w0 + W1);
}
__device__ __forceinline__ void ws_st_fp2(uint32_t* ws, size_t stride, size_t i, int w0, const fp2& a) {
  ws_st_fp(ws, stride, i, w0, a.c0);
  ws_st_fp(ws, stride, i, w0 + W1, a.c1);
}
__device__ __forceinline__ void ws_ld_fp12(fp12& f, const uint32_t* ws, size_t stride, size_t i) {
  ws_ld_fp2(f.c0.a0, ws, stride, i, 0);
  ws_ld_fp2(f.c0.a1, ws, stride, i, W2);
  ws_ld_fp2(f.c0.a2, ws, stride, i, 2 * W2);
  ws_ld_fp2(f.c1.a0, ws, stride, i, 3 * W2);
  ws_ld_fp2(f.c1.a1, ws, stride, i, 4 * W2);
  ws_ld_fp2(f.c1.a2, ws, stride, i, 5 * W2);
}
__device__ __forceinline__ void ws_st_fp12(uint32_t* ws, size_t stride, size_t i, const fp12& f) {
  ws_st_fp2(ws, stride, i, 0, f.c0.a0);
  ws_st_fp2(ws, stride, i, W2, f.c0.a1);
  ws_st_fp2(ws, stride, i, 2 * W2, f.c0.a2);
  ws_st_fp2(ws, stride, i, 3 * W2, f.c1.a0);
  ws_st_fp2(ws, stride, i, 4 * W2, f.c1.a1);
  ws_st_fp2(ws, stride, i, 5 * W2, f.c1.a2);
}
__device__ __forceinline__ void ws_st_pair(uint32_t* ws, size_t stride, size_t i, int slot, const g1_aff& p, const g2_aff& q) {
  const int w0 = slot * 3 * W2;
  ws_st_fp(ws, stride, i, w0, p.x);
  ws_st_fp(ws, stride, i, w0 + W1, p.y);
  ws_st_fp2(ws, stride, i, w0 + W2, q.x);
  ws_st_fp2(ws, stride, i, w0 + 2 * W2, q.y);
}
__device__ __forceinline__ void ws_ld_pair(g1_aff& p, g2_aff& q, const uint32_t* ws, size_t stride, size_t i, int slot) {
  const int w0 = slot * 3 * W2;
  ws_ld_fp(p.x, ws, stride, i, w0);
  ws_ld_fp(p.y, ws, stride, i, w0 + W1);
  ws_ld_fp2(q.x, ws, stride, i, w0 + W2);
  ws_ld_fp2(q.y, ws, stride, i, w0 + 2 * W2);
  p.inf = false;
  q.inf = false;
}

// ---- caller-format point loads (array of structs, one struct per item; blst Montgomery words, 12 per Fp)
__device__ __forceinline__ void fp2_from_raw(fp2& r, const uint32_t* w) {
  fp_from_raw(r.c0, w);
  fp_from_raw(r.c1, w + 12);
}
__device__ __forceinline__ void fp2_to_raw(uint32_t* w, const fp2& a) {
  fp_to_raw(w, a.c0);
  fp_to_raw(w + 12, a.c1);
}
__device__ __forceinline__ bool words_all_zero(const uint32_t* w, int n) {
  uint32_t o = 0;
  for (int k = 0; k < n; k++) o |= w[k];
  return o == 0;
}
__device__ __forceinline__ void load_g1_pt(g1_jac& p, const uint8_t* base, size_t i, int fmt) {
  if (fmt == 0) {
    const uint32_t* w = (const uint32_t*)(base + i * 144);
    fp_from_raw(p.x, w);
    fp_from_raw(p.y, w + 12);
    fp_from_raw(p.z, w + 24);
  } else {
    const uint32_t* w = (const uint32_t*)(base + i * 96);
    if (words_all_zero(w, 24)) {
      jac_set_inf(p);
    } else {
      fp_from_raw(p.x, w);
      fp_from_raw(p.y, w + 12);
      fp_one(p.z);
    }
  }
}
__device__ __forceinline__ void load_g2_pt(g2_jac& p, const uint8_t* base, size_t i, int fmt) {
  if (fmt == 0) {
    const uint32_t* w = (const uint32_t*)(base + i * 288);
    fp2_from_raw(p.x, w);
    fp2_from_raw(p.y, w + 24);
    fp2_from_raw(p.z, w + 48);
  } else {
    const uint32_t* w = (const uint32_t*)(base + i * 192);
    if (words_all_zero(w, 48)) {
      jac_set_inf(p);
    } else {
      fp2_from_raw(p.x, w);
      fp2_from_raw(p.y, w + 24);
      fp2_one(p.z);
    }
  }
}
// the Z coordinate alone (1 for a finite affine point): what the identity checks need
__device__ __forceinline__ void load_g1_z(fp& z, const uint8_t* base, size_t i, int fmt) {
  if (fmt == 0) {
    fp_from_raw(z, (const uint32_t*)(base + i * 144) + 24);
  } else if (words_all_zero((const uint32_t*)(base + i * 96), 24)) {
    fp_zero(z);
  } else {
    fp_one(z);
  }
}
__device__ __forceinline__ void load_g2_z(fp2& z, const uint8_t* base, size_t i, int fmt) {
  if (fmt == 0) {
    fp2_from_raw(z, (const uint32_t*)(base + i * 288) + 48);
  } else if (words_all_zero((const uint32_t*)(base + i * 192), 48)) {
    fp2_zero(z);
  } else {
    fp2_one(z);
  }
}
__device__ __forceinline__ void store_g1_pt(uint8_t* base, size_t i, const g1_jac& p) {
  uint32_t* w = (uint32_t*)(base + i * 144);
  fp_to_raw(w, p.x);
  fp_to_raw(w + 12, p.y);
  fp_to_raw(w + 24, p.z);
}
__device__ __forceinline__ void store_g2_pt(uint8_t* base, size_t i, const g2_jac& p) {
  uint32_t* w = (uint32_t*)(base + i * 288);
  fp2_to_raw(w, p.x);
  fp2_to_raw(w + 24, p.y);
  fp2_to_raw(w + 48, p.z);
}

// ---- lane-split G2 points (jac<hfp2>, tower_split.cuh): each lane of a pair converts and keeps its own component
// (real / imaginary) of every coordinate of a caller-format point
__device__ __forceinline__ void ld_g2s(jac<hfp2>& p, const uint8_t* base, size_t i) {
  const uint32_t* w = (const uint32_t*)(base + i * 288) + (lane_hi() ? 12 : 0);
  fp_from_raw(p.x.v, w);
  fp_from_raw(p.y.v, w + 24);
  fp_from_raw(p.z.v, w + 48);
}
__device__ __forceinline__ void st_g2s(uint8_t* base, size_t i, const jac<hfp2>& p) {
  uint32_t* w = (uint32_t*)(base + i * 288) + (lane_hi() ? 12 : 0);
  fp_to_raw(w, p.x.v);
  fp_to_raw(w + 24, p.y.v);
  fp_to_raw(w + 48, p.z.v);
}
__device__ __forceinline__ void ld_g2s_fmt(jac<hfp2>& p, const uint8_t* base, size_t i, int fmt) {
  if (fmt == 0) {
    ld_g2s(p, base, i);
  } else {
    const uint32_t* w0 = (const uint32_t*)(base + i * 192);
    if (words_all_zero(w0, 48)) {
      jac_set_inf(p);
    } else {
      const uint32_t* w = w0 + (lane_hi() ? 12 : 0);
      fp_from_raw(p.x.v, w);
      fp_from_raw(p.y.v, w + 24);
      fp2_one(p.z);
    }
  }
}


// scalars are taken modulo the group order r (blsgpu.h): a reference Scalar is always canonical, any other 256-bit value is
// reduced here so that the windowed path (255 bits of windows) and the double-and-add path agree on every input
__device__ __forceinline__ void scalar_load_mod_r(uint32_t v[8], const uint8_t* scalars, size_t i) {
  const uint32_t* k = (const uint32_t*)(scalars + 32 * i);
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = k[j];
  const uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  for (int round = 0; round < 3; round++) {   // 2^256 < 2.3 r
    uint32_t d[8];
    uint64_t bw = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t t = (uint64_t)v[j] - R[j] - bw;
      d[j] = (uint32_t)t;
      bw = (t >> 63) & 1;
    }
    if (!bw) {
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = d[j];
    }
  }
}

// ---- kernel prototypes (each kernel is defined in exactly one translation unit, see the BLS_TU_* sections)
template <int SG>
__global__ void k_prepare(size_t n, const uint8_t* pks, const uint8_t* sigs, int fmt, int aug, const uint8_t* msgs,
                          const uint64_t* offs, int single_msg, dst_arg dst, uint32_t* pairs, int32_t* status, int pre_status, int two_lanes);
__global__ void k_miller2s(size_t n, const uint32_t* pairs, const int32_t* status, uint32_t* fws, int fixed_g2);
__global__ void k_lines2s(size_t n, size_t first, size_t count, const uint32_t* pairs, const int32_t* status, uint32_t* lines, uint32_t* lines3, size_t lanes, int fixed_g2, int pass);
__global__ void k_millerf2s(size_t n, size_t first, size_t count, const int32_t* status, const uint32_t* lines, size_t lanes, uint32_t* fws);
__global__ void k_finalexp2s(size_t n, size_t first, size_t count, const uint32_t* fws, uint32_t* vp, size_t lanes, int32_t* status);
__global__ void k_finalexp_seg(int seg, size_t n, size_t first, size_t count, const uint32_t* fws, uint32_t* vp, size_t lanes, int32_t* status);
__global__ void k_cyc_run4(size_t count, uint32_t* vp, size_t lanes, const int32_t* status, size_t first);
__global__ void k_linesp(size_t mm, size_t half, size_t stride, const uint32_t* pairs, const int32_t* bad, uint32_t* lines, uint32_t* lines3, size_t lanes, size_t first_v, size_t count_v, int pass);
__global__ void k_linesp4(size_t cnt, size_t stride, const uint32_t* pairs, const int32_t* bad, uint32_t* lines3, size_t lanes);
__global__ void k_lines_fixed(size_t stride, const uint32_t* pairs, const int32_t* bad, uint32_t* lines3, size_t lanes, int fixed_g2);
__global__ void k_millerfp(size_t count_v, size_t q, int group, const uint32_t* lines, size_t lanes, uint32_t* fws, size_t stride, size_t out0);
__global__ void k_millerfp3(size_t count, size_t q, int group, const int32_t* bad, const uint32_t* lines3, size_t lanes, uint32_t* fws, size_t stride, size_t out0);
__global__ void k_line_quad(size_t count, size_t q, const int32_t* bad, const uint32_t* lines3, size_t lanes, uint32_t* fout, size_t sout, size_t rout, size_t oout);
__global__ void k_f12_fold4(size_t qin, size_t qout, int fan, const uint32_t* fin, size_t sin, size_t rin, uint32_t* fout, size_t sout, size_t rout, size_t oout);
__global__ void k_finalexps(size_t n, const uint32_t* fws, int32_t* status);
__global__ void k_miller1s(size_t n, size_t stride, const uint32_t* pairs, const int32_t* skip, uint32_t* fws);
__global__ void k_finalexp_ones(const uint32_t* fws, size_t stride, int32_t* verdict);
// wave-cooperative variants (coop.cuh): one 64-lane wave per item
__global__ void k_pairing_coop(size_t n, const uint32_t* pairs, int32_t* status, int fixed_g2);
__global__ void k_finalexp_coop(const uint32_t* fws, size_t stride, int32_t* verdict);
template <int SG>
__global__ void k_prepare_agg(size_t n, const uint8_t* pks, const uint8_t* sig, int fmt, int aug, const uint8_t* msgs,
                              const uint64_t* offs, dst_arg dst, uint32_t* pairs, int32_t* bad, int two_lanes, int has_sig);
__global__ void k_pairs_to_affine(size_t n, const uint8_t* g1s, const uint8_t* g2s, int fmt, uint32_t* pairs, int32_t* skip);
__global__ void k_pairs2_to_affine(size_t n, const uint8_t* g1a, const uint8_t* g2a, const uint8_t* g1b, const uint8_t* g2b, int fmt,
                                   uint32_t* pairs, int32_t* status);
__global__ void k_status_to_flag(size_t n, int32_t* status);
template <int SG>
__global__ void k_prepare_hashed(size_t n, const uint8_t* pks, const uint8_t* sigs, const uint8_t* hashes, uint32_t* pairs, int32_t* status, int fmt);
template <int SG>
__global__ void k_prepare_proof(size_t n, const uint8_t* commitments, const uint8_t* proofs, const uint8_t* pks, const uint8_t* ys,
                                int fmt, const uint8_t* msgs, const uint64_t* offs, dst_arg dst, uint32_t* pairs, int32_t* status);
__global__ void k_f12_fold(size_t m, size_t half, uint32_t* fws, size_t stride);
// grouped (random-linear-combination) verification, blsgpu_verify_batch_grouped: see the kernels for the layout
#define GROUPED_ITEMS 8            // items per group; a group is GROUPED_ITEMS + 1 pairs = three Miller loops of three pairs
template <int SG>
__global__ void k_prepare_grouped(size_t n, const uint8_t* pks, const uint8_t* sigs, int fmt, int aug, const uint8_t* msgs, const uint64_t* offs,
                                  dst_arg dst, uint64_t seed, size_t ng, uint32_t* pairs, int32_t* skip, uint8_t* scaled_sigs, int32_t* status);
__global__ void k_group_sigsum(size_t ng, size_t n, const uint8_t* scaled_sigs, uint32_t* pairs, int32_t* skip);
__global__ void k_f12_mul3(size_t ng, const uint32_t* fin, size_t stride_in, uint32_t* fout);
__global__ void k_f12_import(size_t n, const uint8_t* src, uint32_t* fws, size_t stride);
__global__ void k_f12_export(const uint32_t* fws, size_t stride, uint8_t* dst);
__global__ void k_hash_to_g1(size_t n, const uint8_t* msgs, const uint64_t* offs, dst_arg dst, uint8_t* out, int two_lanes);
__global__ void k_hash_to_g2(size_t n, const uint8_t* msgs, const uint64_t* offs, dst_arg dst, uint8_t* out, int two_lanes);
template <int G, int WITH_SCALARS>
__global__ void k_accumulate(size_t n, const uint8_t* pts, int fmt, const uint8_t* scalars, const uint32_t* perm,
                             uint8_t* partials, size_t T);
__global__ void k_accumulate_g2s(size_t n, const uint8_t* pts, int fmt, uint8_t* partials, size_t T);
template <int G>
__global__ void k_point_fold(size_t m, size_t half, uint8_t* partials);
__global__ void k_point_fold_g2s(size_t m, size_t half, uint8_t* partials);
template <int G>
__global__ void k_compress(size_t n, const uint8_t* pts, int fmt, int legacy, uint8_t* out);
template <int SG>
__global__ void k_sign(size_t n, const uint8_t* sks, int aug, const uint8_t* msgs, const uint64_t* offs, dst_arg dst,
                       uint8_t* out_pks, uint8_t* out_sigs);
// Pippenger multi-scalar multiplication (bucket method), see the BLS_TU_MSM section
__global__ void k_msm_count(size_t n, const uint8_t* scalars, int c, int W, int clast, uint32_t* cnt);
__global__ void k_msm_fill(size_t n, const uint8_t* scalars, int c, int W, int clast, const uint32_t* off, uint32_t* cursor, uint32_t* idx);
template <int G>
__global__ void k_msm_bucket(size_t nb, const uint8_t* pts, int fmt, const uint32_t* perm, const uint32_t* cnt, const uint32_t* off,
                             const uint32_t* idx, uint8_t* sums);
template <int G>
__global__ void k_msm_chunk(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials);
__global__ void k_msm_chunk_g2s(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials);
__global__ void k_msm_chunk_g1p(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials);
__global__ void k_msm_chunk_g2q(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials);
__global__ void k_msm_bucket_g2s(size_t nb, const uint8_t* pts, int fmt, const uint32_t* perm, const uint32_t* cnt, const uint32_t* off,
                                 const uint32_t* idx, uint8_t* sums);
template <int G>
__global__ void k_normalize(uint8_t* pt);
// second-generation MSM (msm2.cuh): endomorphism split, signed digits, mixed additions
#define MSM2_E(G) ((G) == 1 ? 2 : 4)               // images per point
#define MSM2_AFF_WORDS(G) ((G) == 1 ? 2 * FP_NL : 4 * FP_NL)   // words per affine entry of the workspace
template <int G>
__global__ void k_msm2_prep(size_t n, const uint8_t* pts, int fmt, const uint32_t* perm, uint32_t* affws, uint8_t* inf);
template <int G>
__global__ void k_msm2_tables(size_t n, const uint32_t* affws, const uint8_t* inf, int W, uint32_t* tab, uint32_t* jt);
template <int G>
__global__ void k_msm2_count(size_t n, const uint8_t* scalars, const uint8_t* inf, int W, uint32_t* cnt, uint64_t* subs);
template <int G>
__global__ void k_msm2_fill(size_t n, const uint64_t* subs, const uint8_t* inf, int W, const uint32_t* off, uint32_t* cursor, uint32_t* idx);
__global__ void k_msm2_merge_g1(size_t nb, int Q, uint8_t* sums);
__global__ void k_msm2_merge_g2s(size_t nb, int Q, uint8_t* sums);
__global__ void k_msm2_bucket_g1(size_t nb, int Q, const uint32_t* affws, const uint32_t* cnt, const uint32_t* off, const uint32_t* idx, uint8_t* sums, int tabW);
__global__ void k_msm2_bucket_g2s(size_t nb, int Q, const uint32_t* affws, const uint32_t* cnt, const uint32_t* off, const uint32_t* idx, uint8_t* sums, int tabW);
__global__ void k_msm2_chunk_g1p(int W, int CH, int stride, const uint8_t* sums, uint8_t* partials, int weighted);
__global__ void k_msm2_chunk_g2q(int W, int CH, int stride, const uint8_t* sums, uint8_t* partials, int weighted);
// wire bytes (48/96 B, modern or legacy header) -> RAW_PROJ with the checks of from_compressed; status[i] = 0 / 7 / 8.
// keep != 0: leave a non-zero status[i] that is already there (first error wins when keys and signatures are decoded)
template <int G>
__global__ void k_decompress(size_t n, const uint8_t* bytes, int legacy, uint8_t* out, int32_t* status, int keep);

#include "util_kernels.cuh"
__global__ void k_wide_mul_test(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int reps);
// single-verification latency path: Miller loop + easy part per item on one wave (coop.cuh), hard part on the row-wide engine
#include "wide_rows.cuh"
#define WIDE_BLOCK 256            // threads of the wide field layer's kernels (hash, multiplier self-test)
#define WIDE_ENGINE_BLOCK (16 * WIDE_TABLE_ROWS)   // threads of the engine kernels: one DPP row per product of a sub-round
#define WIDE_EASY_WORDS (12 * 16)     // f^((p^6-1)(p^2+1)) of one item in the engine's value layout
__global__ void k_pairing_coop_easy(size_t n, const uint32_t* pairs, const int32_t* status, int fixed_g2, uint32_t* easy);
__global__ void k_finalexp_wide(size_t n, const uint32_t* easy, int32_t* status);
__global__ void k_finalexp_wide_ws(const uint32_t* fws, size_t stride, int32_t* verdict);
__global__ void k_f12_tree_wide(size_t m, const uint32_t* fin, size_t sin, uint32_t* fout, size_t sout);
__global__ void k_f12_tree_seg(size_t qin, const uint32_t* fin, size_t sin, size_t rin, uint32_t* fout, size_t sout, size_t rout);
__global__ void k_f12_horner_wide(const uint32_t* fin, size_t sin, uint32_t* fout, size_t sout);
__global__ void k_wide_prog_test(const uint32_t* prog, int len, int reps, const uint8_t* fin, uint8_t* tout);
bool wide_prog_is_fp12(const uint32_t* prog, size_t len);   // host: what k_wide_prog_test may be given
__global__ void k_pairing_wide(size_t n, const uint32_t* pairs, int32_t* status, int fixed_g2);
__global__ void k_hash_to_g1_wide(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst, uint8_t* out, uint32_t* rec);
// the same with one WORKGROUP per message: wave 0 hashes up to the sum of the two mapped points, then all four waves clear the
// cofactor on the engine (program G1_HASH_TAIL); for up to 128 messages
__global__ void k_hash_to_g1_engine(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst, uint8_t* out, uint32_t* rec);
// The same check cut where its inputs become known (csrc/wide_tables.cuh programs PRE_LINES, PRE_F1 / PRE_F1G, POST).  The pairs
// are (P0, Q0) (P1, Q1) = (H(m), key) (signature, -g2) for Bls12381G1Impl and (key, H(m)) (-g1, signature) for Bls12381G2Impl.
// Per item a RECORD of engine values (16 words each) in global memory carries the operands and what the early parts hand over:
#define WREC_P0 0        // pair 0's G1 point, Jacobian X Y Z      (G1Impl: the message point BEFORE its cofactor clearing, k_hash_to_g1_*;  G2Impl: the key, k_prepare_keys part 2)
#define WREC_P1 3        // G1Impl: the signature, Jacobian X Y Z  (k_prepare_keys, part 1)
#define WREC_Q0 6        // pair 0's G2 point, Jacobian in Fp2: 6 values (G1Impl: the key, part 2;  G2Impl: H(m), part 4)
#define WREC_F1 12       // G1Impl: Miller function of (signature, -[c] g2), the pair that balances the uncleared P0 (k_pairing_pre, part 1)
#define WREC_L 24        // pair 0's 68 unscaled lines, 6 values each (k_pairing_pre, part 0)
#define WREC_Q1 (24 + 6 * 68)   // Bls12381G2Impl only: pair 1's G2 point (the signature), Jacobian: 6 values
#define WREC_F2 (WREC_Q1 + 6)    // the split Miller loop's hand-over slots (k_pairing_post2, k_pairing_stream): the later parts' Fp12 values, 12 each
#define WREC_F2_SLOTS 2
#define WREC_VALUES (WREC_F2 + 12 * WREC_F2_SLOTS)
#define WREC_WORDS (16 * WREC_VALUES)
template <int SG>
__global__ void k_prepare_keys(size_t n, const uint8_t* pks, const uint8_t* sigs, const uint8_t* hashes, int fmt, int parts, uint32_t* rec, int32_t* status);
__global__ void k_pairing_pre(size_t n, uint32_t* rec, const int32_t* status, int first_part, int second_part);   // grid (n, 1 or 2): part 0 = pair 0's lines, 1 = F1 (G1Impl), 2 = F1 (G2Impl)
__global__ void k_pairing_post(size_t n, const uint32_t* rec, int32_t* status);
// part 0 and k_pairing_post side by side, for the checks whose lines cannot be had early (the summed key of MultiSignature::verify /
// verify_secure, Bls12381G2Impl's H(m)): grid (n, 3), workgroup (i, 0) derives item i's lines (program PRE_LINES_S) and hands them
// over through the record a few steps at a time (tools/gen_wide_tables.py STREAM_BOUNDS); workgroups (i, 1) and (i, 2) run the Miller
// loop on them as they arrive, split as in k_pairing_post2 below (POST_LO_S: the last 27 iterations from 1; POST_HI_S: the rest of POST).  flags: WSTREAM_FLAGS words
// per item in a buffer that ONLY these kernels write; epoch: a value no earlier launch on that buffer used (never 0).  n <= WSTREAM_MAX_ITEMS:
// the workgroups of an item must be resident together.
#define WSTREAM_FLAGS 32              // one per four line steps: a chunk's flag is its first step / 4
#define WSTREAM_MAX_ITEMS 64             // k_pairing_stream: three workgroups per item
#define WPOST2_MAX_ITEMS 128            // k_pairing_post2: two per item (the flag buffer has room for this many items)
#define WSTREAM_SPIN_LIMIT (1u << 21)      // polls (~1 us each) before the consumer gives up: status BLS_ERR_STREAM_TIMEOUT
#define BLS_ERR_STREAM_TIMEOUT (-2)        // = BLSGPU_E_HIP: a device-side failure, not a verdict
__global__ void k_pairing_stream(size_t n, uint32_t* rec, int32_t* status, uint32_t* flags, uint32_t epoch);
// k_pairing_post with its Miller loop on two or three workgroups: grid (n, 2) -- programs POST_LO / POST_HI, the last 41 iterations
// from 1 beside the first 22 and 41 squarings -- or grid (n, 3) -- POST3_LO / POST3_MID / POST3_HI: 36 iterations, 18 and 36 squarings, 9
// and 54 --; the workgroups with the lower block indices hand their accumulators to the last one through the record; same flags
// and epoch as k_pairing_stream; n <= WPOST2_MAX_ITEMS (grid (n, 3): WSTREAM_MAX_ITEMS)
__global__ void k_pairing_post2(size_t n, uint32_t* rec, int32_t* status, uint32_t* flags, uint32_t epoch);
// the last levels of a point sum on the engine: workgroup b <- the sum of points [16 b, 16 b + 16) (RAW_PROJ in and out)
template <int G>
__global__ void k_point_tree_wide(size_t m, const uint8_t* in, uint8_t* out);
// the cofactor clearing of hash-to-G2 on the engine: pts[i] <- h_eff pts[i] (RAW_PROJ, in place), one workgroup per point
__global__ void k_g2_clear_wide(size_t n, uint8_t* pts);
// hash-to-G2 with one WORKGROUP per message: wave 0 expands the message and runs the two SSWU maps and isogenies on two DPP rows
// (row-wide Fp2, csrc/wide_fp2.cuh), then all four waves add the two points and clear the cofactor on the engine (program
// G2_HASH_TAIL).  out: RAW_PROJ (Jacobian).  single_msg bit 0: every item hashes message 0.  For up to 128 messages.
__global__ void k_hash_to_g2_engine(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst, uint8_t* out, uint32_t* rec);   // out / rec: either may be null; rec: the cut check's record (WREC_Q0)
// the same for more messages as two launches: the maps with one WAVE per message (pts: 12 engine values = 768 bytes per message),
// then the engine program with one workgroup per message
__global__ void k_hash_to_g2_maps(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst, uint32_t* pts);
__global__ void k_g2_hash_tail_wide(size_t n, const uint32_t* pts, uint8_t* out, uint32_t* rec);

#if defined(BLS_TU_PREPARE1) || defined(BLS_TU_PREPARE2)
// =====================================================================================================
// verify_batch stage 1: identity checks, to-affine, Aug prefix, hash-to-curve  ->  two affine pairs per item.
// single_msg != 0: every item uses message [offs[0], offs[1]) (multi_verify / verify_secure tail).
template <int SG>
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_prepare(size_t n, const uint8_t* pks, const uint8_t* sigs, int fmt, int aug,
                                                     const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst,
                                                     uint32_t* pairs, int32_t* status, int pre_status, int two_lanes) {
  // two_lanes bit 0 (small batches, single-verify tails): two adjacent lanes per item run the two SSWU maps of the hash side by
  // side (h2c.cuh) and everything else redundantly with identical operands; lane 0 of the pair stores.
  // bit 1 (SG == 1): the message point stays uncleared and pair 1 becomes (sig, -[c] g2) (verify.cuh prepare_g1impl): the
  // pairing that follows must take the line table of THAT point (fixed_g2 = 2)
  // (Bls12381G2Impl is only ever launched with two lanes per item: a compile-time fact there, so that the one-lane form of hash_to_g2 --
  // 5 KB of frame that no lane ever touched -- is not part of k_prepare<2>)
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool tl = SG == 2 || (two_lanes & 1) != 0;
  const size_t i = tl ? gid >> 1 : gid;
  const int lane2 = tl ? (int)(gid & 1) : -1;
  if (i >= n) return;
  if (pre_status && status[i] != BLS_OK) return;   // the item already failed to decode
  size_t mi = single_msg ? 0 : i;
  const uint8_t* m = msgs + offs[mi];
  uint32_t mlen = (uint32_t)(offs[mi + 1] - offs[mi]);
  g1_aff P[2];
  g2_aff Q[2];
  int st;
  if (SG == 1 && !aug) {
    // Key and signature are not needed before the hash is through (nothing of them enters it): only their Z coordinates are read
    // first, for the identity checks in the reference's order, and the points themselves after the hash -- held from the start they
    // were 126 dwords parked in scratch across the whole hash (a store and a load each, at scratch latency)
    fp zs;
    fp2 zp;
    load_g1_z(zs, sigs, i, fmt);
    load_g2_z(zp, pks, i, fmt);
    if (fp_is_zero(zs)) {
      st = BLS_ERR_SIG_IDENTITY;
    } else if (fp2_is_zero(zp)) {
      st = BLS_ERR_PK_IDENTITY;
    } else {
      const bool no_clear = (two_lanes & 2) != 0;
      g1_jac h;
      hash_to_g1(h, nullptr, 0, m, mlen, dst.b, dst.len, lane2, no_clear);
      g2_jac pk;
      g1_jac sig;
      load_g2_pt(pk, pks, i, fmt);
      load_g1_pt(sig, sigs, i, fmt);
      prepare_g1impl_tail(P, Q, pk, sig, h);
      if (no_clear) g2_negc_gen(Q[1]);
      else g2_neg_gen(Q[1]);
      st = BLS_OK;
    }
  } else if (SG == 1) {
    g2_jac pk;
    g1_jac sig;
    load_g2_pt(pk, pks, i, fmt);
    load_g1_pt(sig, sigs, i, fmt);
    st = prepare_g1impl(P, Q, pk, sig, aug, m, mlen, dst.b, dst.len, lane2, (two_lanes & 2) != 0);
  } else {
    g1_jac pk;
    g2_jac sig;
    load_g1_pt(pk, pks, i, fmt);
    load_g2_pt(sig, sigs, i, fmt);
    st = prepare_g2impl(P, Q, pk, sig, aug, m, mlen, dst.b, dst.len, lane2);
  }
  if (lane2 > 0) return;
  status[i] = st;
  if (st != BLS_OK) return;
  ws_st_pair(pairs, n, i, 0, P[0], Q[0]);
  ws_st_pair(pairs, n, i, 1, P[1], Q[1]);
}
// Grouped verification (opt-in, blsgpu_verify_batch_grouped): GROUPED_ITEMS items share ONE final exponentiation.  With per-item
// scalars r_i a group is accepted iff
//     prod_i e(r_i H(m_i), pk_i) * e(sum_i r_i sig_i, -g2) == 1,
// which holds for valid signatures; a group that fails is re-verified item by item, so a valid item is never rejected and the
// statuses are those of blsgpu_verify_batch.  The scalars are DERIVED FROM THE GROUP'S OWN INPUTS (Fiat-Shamir; round 3, after
// the advisor's finding on seed-only scalars: with r_a, r_b known in advance sig_a + [r_b] D and sig_b - [r_a] D cancel in the
// combined check): d_i = SHA-256(tag || seed || n || i || pk_i || sig_i || |m_i| || m_i) over the caller's bytes, D = SHA-256(d of the
// group's eight slots, zero for slots past n), r_i = the first 128 bits of SHA-256(D || i) (1 if they are all zero; 128 bits since round 4:
// the advisor's offline grinding against 64-bit scalars with a known seed took 2^64 hash evaluations).  Whoever
// chooses the inputs of a group learns its scalars only after all of them are fixed, so making an invalid item pass takes about
// 2^128 hash evaluations per group; the caller's seed is mixed in as optional extra entropy and needs no secrecy.
// This kernel: the per-item checks and hash of k_prepare, then A_i = r_i H(m_i) (affine, with the key: one pair of the
// group's Miller loops) and B_i = r_i sig_i (Jacobian, for k_group_sigsum).  Pair slots: item i sits at (i / 8) + (i % 8) ng
// of a workspace of 9 ng one-pair items, so that k_miller1s (items g, g + q, g + 2q per loop, q = 3 ng) leaves the three
// partial products of group c at c, c + ng, c + 2 ng.  skip[] arrives all ones.
__device__ __forceinline__ void sha256_u64(sha256_ctx& c, uint64_t x) {
  for (int k = 7; k >= 0; k--) sha256_byte(c, (uint8_t)(x >> (8 * k)));
}
// the scalar of item i; every lane of the wave calls this (lanes past n contribute zero digests), groups are aligned octets of lanes
struct u128_pair { uint64_t hi, lo; };
__device__ __noinline__ u128_pair grouped_scalar(uint64_t seed, size_t n, size_t i, const uint8_t* pk, uint32_t pk_len, const uint8_t* sig, uint32_t sig_len,
                                              const uint8_t* m, uint32_t mlen) {
  static_assert(GROUPED_ITEMS == 8 && BLS_BLOCK % GROUPED_ITEMS == 0, "a group is an aligned octet of lanes");
  uint8_t d[32];
  sha256_ctx c;
  if (i < n) {
    const char tag[] = "blsgpu-grouped-v3";
    sha256_init(c);
    sha256_update(c, (const uint8_t*)tag, sizeof tag - 1);
    sha256_u64(c, seed);
    sha256_u64(c, (uint64_t)n);
    sha256_u64(c, (uint64_t)i);
    sha256_update(c, pk, pk_len);
    sha256_update(c, sig, sig_len);
    sha256_u64(c, mlen);
    sha256_update(c, m, mlen);
    sha256_final(c, d);
  } else {
    for (int k = 0; k < 32; k++) d[k] = 0;
  }
  uint32_t w[8];
  for (int k = 0; k < 8; k++) w[k] = ((uint32_t)d[4 * k] << 24) | ((uint32_t)d[4 * k + 1] << 16) | ((uint32_t)d[4 * k + 2] << 8) | d[4 * k + 3];
  sha256_init(c);
  for (int j = 0; j < GROUPED_ITEMS; j++)
    for (int k = 0; k < 8; k++) {
      const uint32_t x = (uint32_t)__shfl((int)w[k], j, GROUPED_ITEMS);
      sha256_byte(c, (uint8_t)(x >> 24));
      sha256_byte(c, (uint8_t)(x >> 16));
      sha256_byte(c, (uint8_t)(x >> 8));
      sha256_byte(c, (uint8_t)x);
    }
  sha256_final(c, d);
  sha256_init(c);
  sha256_update(c, d, 32);
  sha256_u64(c, (uint64_t)i);
  sha256_final(c, d);
  u128_pair r{0, 0};
  for (int k = 0; k < 8; k++) r.hi = (r.hi << 8) | d[k];
  for (int k = 8; k < 16; k++) r.lo = (r.lo << 8) | d[k];
  if (!(r.hi | r.lo)) r.lo = 1;
  return r;
}
template <int SG>
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_prepare_grouped(size_t n, const uint8_t* pks, const uint8_t* sigs, int fmt, int aug, const uint8_t* msgs,
                                                             const uint64_t* offs, dst_arg dst, uint64_t seed, size_t ng, uint32_t* pairs, int32_t* skip,
                                                             uint8_t* scaled_sigs, int32_t* status) {
  static_assert(SG == 1, "grouped verification is built for Bls12381G1Impl");
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const size_t ii = live ? i : 0;
  const uint8_t* m = msgs + offs[ii];
  const uint32_t mlen = (uint32_t)(offs[ii + 1] - offs[ii]);
  const uint32_t pk_len = fmt == 0 ? 288 : 192, sig_len = fmt == 0 ? 144 : 96;
  const u128_pair r = grouped_scalar(seed, n, i, pks + ii * pk_len, pk_len, sigs + ii * sig_len, sig_len, m, mlen);   // all lanes: the group's digests travel by shuffles
  if (!live) return;
  const size_t slot = i / GROUPED_ITEMS + (i % GROUPED_ITEMS) * ng, stride = (GROUPED_ITEMS + 1) * ng;
  g1_aff P[2];
  g2_aff Q[2];
  g2_jac pk;
  g1_jac sig, a, b;
  load_g2_pt(pk, pks, i, fmt);
  load_g1_pt(sig, sigs, i, fmt);
  const int st = prepare_g1impl(P, Q, pk, sig, aug, m, mlen, dst.b, dst.len, -1, true);   // uncleared message point: the group's last pair balances it
  status[i] = st;
  if (st != BLS_OK) {              // excluded from its group (the slot stays skipped); its status is final
    jac_set_inf(b);
    store_g1_pt(scaled_sigs, i, b);
    return;
  }
  jac_from_aff(a, P[0]);
  jac_mul_u128(a, a, r.hi, r.lo);
  jac_from_aff(b, P[1]);
  jac_mul_u128(b, b, r.hi, r.lo);
  g1_aff A;
  jac_to_aff(A, a);
  ws_st_pair(pairs, stride, slot, 0, A, Q[0]);
  skip[slot] = A.inf ? 1 : 0;
  store_g1_pt(scaled_sigs, i, b);
}
#if defined(BLS_TU_PREPARE1)
template __global__ void k_prepare_grouped<1>(size_t, const uint8_t*, const uint8_t*, int, int, const uint8_t*, const uint64_t*, dst_arg, uint64_t, size_t,
                                              uint32_t*, int32_t*, uint8_t*, int32_t*);
#endif
#if defined(BLS_TU_PREPARE1)
template __global__ void k_prepare<1>(size_t, const uint8_t*, const uint8_t*, int, int, const uint8_t*, const uint64_t*, int, dst_arg, uint32_t*, int32_t*, int, int);
#else
template __global__ void k_prepare<2>(size_t, const uint8_t*, const uint8_t*, int, int, const uint8_t*, const uint64_t*, int, dst_arg, uint32_t*, int32_t*, int, int);
#endif
#endif  // BLS_TU_PREPARE*



#if defined(BLS_TU_AGG1) || defined(BLS_TU_AGG2)
// =====================================================================================================
// aggregate verify / pairing product: one pair per item, then a product tree over the Fp12 values.
// mode 0: (g1s[i], g2s[i]) given by the caller (pairing_product_is_one)
// mode 1: aggregate verify: pk[i] + message i -> pair (H(m_i), pk_i) in the G1-first order of src/helpers.rs;
//         item n (the extra lane) carries (sig, -g).  bad[i] = 1 when pk_i is the identity.
template <int SG>
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_prepare_agg(size_t n, const uint8_t* pks, const uint8_t* sig, int fmt, int aug,
                                                         const uint8_t* msgs, const uint64_t* offs, dst_arg dst, uint32_t* pairs,
                                                         int32_t* bad, int two_lanes, int has_sig) {
  // two_lanes bit 0: as k_prepare (two adjacent lanes per item, the hash's two SSWU maps side by side; lane 0 stores);
  // bit 1 (SG == 1): the hashes stay in E1(Fp), uncleared, and the signature's pair is (sig, -[c] g2) (verify.cuh g2_negc_gen)
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = (two_lanes & 1) ? gid >> 1 : gid;
  const int lane2 = (two_lanes & 1) ? (int)(gid & 1) : -1;
  if (i > n || (i == n && !has_sig)) return;   // has_sig == 0: a shard without the signature pair (sig may be null)
  const size_t stride = n + 1;
  g1_aff P;
  g2_aff Q;
  if (i == n) {  // the signature pair
    if (SG == 1) {
      g1_jac s;
      load_g1_pt(s, sig, 0, fmt);
      const bool inf = jac_is_inf(s);
      if (lane2 <= 0) bad[i] = inf ? 1 : 0;
      if (inf) return;
      jac_to_aff(P, s);
      if (two_lanes & 2) g2_negc_gen(Q);          // the pairs' message points are uncleared: (sig, -[c] g2) balances them
      else g2_neg_gen(Q);
    } else {
      g2_jac s;
      load_g2_pt(s, sig, 0, fmt);
      const bool inf = jac_is_inf(s);
      if (lane2 <= 0) bad[i] = inf ? 1 : 0;
      if (inf) return;
      jac_to_aff(Q, s);
      g1_neg_gen(P);
    }
    if (lane2 <= 0) ws_st_pair(pairs, stride, i, 0, P, Q);
    return;
  }
  const uint8_t* m = msgs + offs[i];
  uint32_t mlen = (uint32_t)(offs[i + 1] - offs[i]);
  if (SG == 1) {
    g2_jac pk;
    load_g2_pt(pk, pks, i, fmt);
    const bool inf = jac_is_inf(pk);
    if (lane2 <= 0) bad[i] = inf ? 1 : 0;
    if (inf) return;
    uint8_t pre[96];
    g1_jac h;
    if (aug) {                               // the key's bytes prefix the message: its affine form comes first
      jac_to_aff(Q, pk);
      g2_compress(pre, Q, false);
      hash_to_g1(h, pre, 96, m, mlen, dst.b, dst.len, lane2, (two_lanes & 2) != 0);
      jac_to_aff(P, h);
    } else {
      hash_to_g1(h, pre, 0, m, mlen, dst.b, dst.len, lane2, (two_lanes & 2) != 0);
      if (jac_is_inf(h)) {                   // cannot happen for a hash output in practice; keep the generic path correct
        jac_to_aff(Q, pk);
        jac_to_aff(P, h);
      } else {
        g1g2_to_aff(P, Q, h, pk);            // one inversion for both affine forms
      }
    }
  } else {
    g1_jac pk;
    load_g1_pt(pk, pks, i, fmt);
    const bool inf = jac_is_inf(pk);
    if (lane2 <= 0) bad[i] = inf ? 1 : 0;
    if (inf) return;
    uint8_t pre[48];
    g2_jac h;
    if (aug) {
      jac_to_aff(P, pk);
      g1_compress(pre, P, false);
      hash_to_g2(h, pre, 48, m, mlen, dst.b, dst.len, lane2);
      jac_to_aff(Q, h);
    } else {
      hash_to_g2(h, pre, 0, m, mlen, dst.b, dst.len, lane2);
      if (jac_is_inf(h)) {
        jac_to_aff(P, pk);
        jac_to_aff(Q, h);
      } else {
        g1g2_to_aff(P, Q, pk, h);
      }
    }
  }
  if (lane2 <= 0) ws_st_pair(pairs, stride, i, 0, P, Q);
}

#if defined(BLS_TU_AGG1)
template __global__ void k_prepare_agg<1>(size_t, const uint8_t*, const uint8_t*, int, int, const uint8_t*, const uint64_t*, dst_arg, uint32_t*, int32_t*, int, int);
#else
template __global__ void k_prepare_agg<2>(size_t, const uint8_t*, const uint8_t*, int, int, const uint8_t*, const uint64_t*, dst_arg, uint32_t*, int32_t*, int, int);
#endif
#endif  // BLS_TU_AGG*

#if defined(BLS_TU_POINTS)
// caller-supplied pairs -> affine workspace; pairs with an identity member are flagged and contribute 1
__global__ void __launch_bounds__(BLS_BLOCK) k_pairs_to_affine(size_t n, const uint8_t* g1s, const uint8_t* g2s, int fmt,
                                                             uint32_t* pairs, int32_t* skip) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_jac a;
  g2_jac b;
  load_g1_pt(a, g1s, i, fmt);
  load_g2_pt(b, g2s, i, fmt);
  skip[i] = (jac_is_inf(a) || jac_is_inf(b)) ? 1 : 0;
  if (skip[i]) return;
  g1_aff P;
  g2_aff Q;
  g1g2_to_aff(P, Q, a, b);
  ws_st_pair(pairs, n, i, 0, P, Q);
}

// n independent two-pair products e(a1, a2) * e(b1, b2) (Pairing::pairing on two pairs, reference src/helpers.rs:41-63):
// both pairs of an item to affine in the layout the two-pair pairing stages read.  A pair with an identity member
// contributes 1, as in the reference's multi_miller_loop: with both pairs trivial the item gets a fixed product that IS one
// (e(g1, -g2) * e(-g1, -g2)); with exactly one trivial pair the product is a single pairing of two non-identity subgroup
// points, which is never one (non-degeneracy), so the verdict is written here and the pairing stages skip the item.
__global__ void __launch_bounds__(BLS_BLOCK) k_pairs2_to_affine(size_t n, const uint8_t* g1a, const uint8_t* g2a, const uint8_t* g1b,
                                                              const uint8_t* g2b, int fmt, uint32_t* pairs, int32_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_jac a1, b1;
  g2_jac a2, b2;
  load_g1_pt(a1, g1a, i, fmt);
  load_g2_pt(a2, g2a, i, fmt);
  load_g1_pt(b1, g1b, i, fmt);
  load_g2_pt(b2, g2b, i, fmt);
  const bool ta = jac_is_inf(a1) || jac_is_inf(a2), tb = jac_is_inf(b1) || jac_is_inf(b2);
  g1_aff P[2];
  g2_aff Q[2];
  if (ta != tb) {
    status[i] = BLS_ERR_INVALID_SIGNATURE;
    return;
  }
  status[i] = BLS_OK;
  if (ta) {
    fp_load(P[0].x, G1_GEN_X);
    fp_load(P[0].y, G1_GEN_Y);
    P[0].inf = false;
    g1_neg_gen(P[1]);
    g2_neg_gen(Q[0]);
    g2_neg_gen(Q[1]);
  } else {
    g1g2_to_aff(P[0], Q[0], a1, a2);
    g1g2_to_aff(P[1], Q[1], b1, b2);
  }
  ws_st_pair(pairs, n, i, 0, P[0], Q[0]);
  ws_st_pair(pairs, n, i, 1, P[1], Q[1]);
}

// core_verify's stage 1 when H(msg) is already known (RAW_PROJ in `hashes`): the single-verify tail of verify_secure hashes
// its message while the host still derives the coefficients, so only the identity checks (signature first, then key:
// reference src/traits/sig_core.rs:126-135) and the shared-inversion conversion to affine pairs are left.
template <int SG>
__global__ void __launch_bounds__(BLS_BLOCK) k_prepare_hashed(size_t n, const uint8_t* pks, const uint8_t* sigs, const uint8_t* hashes,
                                                            uint32_t* pairs, int32_t* status, int fmt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_aff P[2];
  g2_aff Q[2];
  int st = BLS_OK;
  if (SG == 1) {
    g2_jac pk;
    g1_jac sig, h;
    load_g2_pt(pk, pks, i, fmt);
    load_g1_pt(sig, sigs, i, fmt);
    load_g1_pt(h, hashes, i, 0);
    if (jac_is_inf(sig)) st = BLS_ERR_SIG_IDENTITY;
    else if (jac_is_inf(pk)) st = BLS_ERR_PK_IDENTITY;
    else if (jac_is_inf(h)) {
      g1g2_to_aff(P[1], Q[0], sig, pk);
      jac_to_aff(P[0], h);
    } else {
      fp zs = sig.z, zh = h.z, nn;
      fp2_norm_sq(nn, pk.z);
      fp_inv3(zs, zh, nn);
      g1_apply_zinv(P[1], sig, zs);
      g1_apply_zinv(P[0], h, zh);
      g2_apply_ninv(Q[0], pk, nn);
    }
    g2_neg_gen(Q[1]);
  } else {
    g1_jac pk;
    g2_jac sig, h;
    load_g1_pt(pk, pks, i, fmt);
    load_g2_pt(sig, sigs, i, fmt);
    load_g2_pt(h, hashes, i, 0);
    if (jac_is_inf(sig)) st = BLS_ERR_SIG_IDENTITY;
    else if (jac_is_inf(pk)) st = BLS_ERR_PK_IDENTITY;
    else if (jac_is_inf(h)) {
      g1g2_to_aff(P[0], Q[1], pk, sig);
      jac_to_aff(Q[0], h);
    } else {
      fp zp = pk.z, ns, nh;
      fp2_norm_sq(ns, sig.z);
      fp2_norm_sq(nh, h.z);
      fp_inv3(zp, ns, nh);
      g1_apply_zinv(P[0], pk, zp);
      g2_apply_ninv(Q[1], sig, ns);
      g2_apply_ninv(Q[0], h, nh);
    }
    g1_neg_gen(P[1]);
  }
  status[i] = st;
  if (st != BLS_OK) return;
  ws_st_pair(pairs, n, i, 0, P[0], Q[0]);
  ws_st_pair(pairs, n, i, 1, P[1], Q[1]);
}
template __global__ void k_prepare_hashed<1>(size_t, const uint8_t*, const uint8_t*, const uint8_t*, uint32_t*, int32_t*, int);
template __global__ void k_prepare_hashed<2>(size_t, const uint8_t*, const uint8_t*, const uint8_t*, uint32_t*, int32_t*, int);

// The operands of the cut check (k_pairing_pre / k_pairing_post) that do not depend on the message, written as engine values
// into the item's record.  parts & 1: the signature -- identity check (first, reference src/traits/sig_core.rs:126-129);
// parts & 2: the key -- identity check (:130-135, only if the signature passed); parts & 4 (Bls12381G2Impl): H(m) from
// k_hash_to_g2's output.  All are stored as they come, Jacobian: the
// engine's programs evaluate lines at projective G1 points and walk the Miller loop from a projective G2 point
// (tools/gen_wide_tables.py op_lscale, op_padd*), so NO inversion is spent on the way to the pairing.  The two parts may come
// in separate launches (a multi-signature's key is only known after the sum): the signature part runs first and sets status,
// the key part keeps a failure it finds there.
template <int SG>
__global__ void __launch_bounds__(BLS_BLOCK) k_prepare_keys(size_t n, const uint8_t* pks, const uint8_t* sigs, const uint8_t* hashes, int fmt, int parts,
                                                          uint32_t* rec, int32_t* status) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t* r = rec + i * WREC_WORDS;
  auto put = [&](int v, const fp& a) {
    fp t;
    fp_reduce(t, a);
#pragma unroll
    for (int k = 0; k < FP_NL; k++) r[16 * v + k] = (uint32_t)t.l[k];
    r[16 * v + 14] = 0;
    r[16 * v + 15] = 0;
  };
  auto put_g1 = [&](int v, const g1_jac& p) {
    put(v, p.x);
    put(v + 1, p.y);
    put(v + 2, p.z);
  };
  auto put_g2 = [&](int v, const g2_jac& p) {
    put(v, p.x.c0);
    put(v + 1, p.x.c1);
    put(v + 2, p.y.c0);
    put(v + 3, p.y.c1);
    put(v + 4, p.z.c0);
    put(v + 5, p.z.c1);
  };
  int st = (parts & 1) ? BLS_OK : ((parts & 2) ? status[i] : BLS_OK);
  if (SG == 1) {
    if (parts & 1) {
      g1_jac sig;
      load_g1_pt(sig, sigs, i, fmt);
      if (jac_is_inf(sig)) st = BLS_ERR_SIG_IDENTITY;
      put_g1(WREC_P1, sig);
    }
    if (parts & 2) {
      g2_jac pk;
      load_g2_pt(pk, pks, i, fmt);
      if (jac_is_inf(pk) && st == BLS_OK) st = BLS_ERR_PK_IDENTITY;
      put_g2(WREC_Q0, pk);
    }
  } else {
    if (parts & 1) {
      g2_jac sig;
      load_g2_pt(sig, sigs, i, fmt);
      if (jac_is_inf(sig)) st = BLS_ERR_SIG_IDENTITY;
      put_g2(WREC_Q1, sig);
    }
    if (parts & 2) {
      g1_jac pk;
      load_g1_pt(pk, pks, i, fmt);
      if (jac_is_inf(pk) && st == BLS_OK) st = BLS_ERR_PK_IDENTITY;
      put_g1(WREC_P0, pk);
    }
    if (parts & 4) {                     // H(m) in G2, as k_hash_to_g2 left it (RAW_PROJ); touches no status
      g2_jac h;
      load_g2_pt(h, hashes, i, 0);
      put_g2(WREC_Q0, h);
    }
  }
  if (parts & 3) status[i] = st;
}
template __global__ void k_prepare_keys<1>(size_t, const uint8_t*, const uint8_t*, const uint8_t*, int, int, uint32_t*, int32_t*);
template __global__ void k_prepare_keys<2>(size_t, const uint8_t*, const uint8_t*, const uint8_t*, int, int, uint32_t*, int32_t*);

// status (0 = product is one) -> is_one flag (1 / 0)
__global__ void __launch_bounds__(BLS_BLOCK) k_status_to_flag(size_t n, int32_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) status[i] = status[i] == BLS_OK ? 1 : 0;
}

#endif  // BLS_TU_POINTS (pairs_to_affine)


#if defined(BLS_TU_FINALEXP)
// product tree step: f[i] *= f[i + half] for i + half < m   (stride = workspace stride)
__global__ void __launch_bounds__(BLS_BLOCK) k_f12_fold(size_t m, size_t half, uint32_t* fws, size_t stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + half >= m || i >= half) return;
  fp12 a, b;
  ws_ld_fp12(a, fws, stride, i);
  ws_ld_fp12(b, fws, stride, i + half);
  fp12_mul(a, a, b);
  ws_st_fp12(fws, stride, i, a);
}

// grouped verification: the three partial Miller products of group c (at c, c + ng, c + 2 ng of the input workspace) -> out[c]
__global__ void __launch_bounds__(BLS_BLOCK) k_f12_mul3(size_t ng, const uint32_t* fin, size_t stride_in, uint32_t* fout) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ng) return;
  fp12 a, b;
  ws_ld_fp12(a, fin, stride_in, c);
  ws_ld_fp12(b, fin, stride_in, c + ng);
  fp12_mul(a, a, b);
  ws_ld_fp12(b, fin, stride_in, c + 2 * ng);
  fp12_mul(a, a, b);
  ws_st_fp12(fout, ng, c, a);
}

// Fp12 partial products cross the C ABI as 576-byte records (c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2; each Fp2 =
// c0 then c1; Montgomery limbs): array-of-records <-> word-major workspace
__global__ void __launch_bounds__(BLS_BLOCK) k_f12_import(size_t n, const uint8_t* src, uint32_t* fws, size_t stride) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* w = (const uint32_t*)(src + i * 576);
  for (int k = 0; k < 12; k++) {
    fp t;
    fp_from_raw(t, w + 12 * k);
    ws_st_fp(fws, stride, i, W1 * k, t);
  }
}
__global__ void __launch_bounds__(BLS_BLOCK) k_f12_export(const uint32_t* fws, size_t stride, uint8_t* dst) {
  if (blockIdx.x != 0 || threadIdx.x >= 12) return;
  fp t;
  ws_ld_fp(t, fws, stride, 0, W1 * threadIdx.x);
  fp_to_raw((uint32_t*)dst + 12 * threadIdx.x, t);
}
#endif  // BLS_TU_FINALEXP

#if defined(BLS_TU_POINTS)
#include "msm2.cuh"     // jac_madd
// =====================================================================================================
// hash_to_point batches
// two_lanes: two adjacent lanes per message run the two SSWU maps side by side (and, for G2, the cofactor clearing on the
// lane-split tower), as k_prepare does: the latency mode for the single message of a verify_secure tail
__global__ void __launch_bounds__(BLS_BLOCK) k_hash_to_g1(size_t n, const uint8_t* msgs, const uint64_t* offs, dst_arg dst, uint8_t* out, int two_lanes) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = two_lanes ? gid >> 1 : gid;
  const int lane2 = two_lanes ? (int)(gid & 1) : -1;
  if (i >= n) return;
  g1_jac h;
  hash_to_g1(h, nullptr, 0, msgs + offs[i], (uint32_t)(offs[i + 1] - offs[i]), dst.b, dst.len, lane2);
  if (lane2 <= 0) store_g1_pt(out, i, h);
}
// two_lanes bit 1 (value 2): stop before the cofactor clearing (k_g2_clear_wide does it on the row-wide engine)
__global__ void __launch_bounds__(BLS_BLOCK) k_hash_to_g2(size_t n, const uint8_t* msgs, const uint64_t* offs, dst_arg dst, uint8_t* out, int two_lanes) {
  const bool no_clear = (two_lanes & 2) != 0;
  two_lanes &= 1;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = two_lanes ? gid >> 1 : gid;
  const int lane2 = two_lanes ? (int)(gid & 1) : -1;
  if (i >= n) return;
  g2_jac h;
  hash_to_g2(h, nullptr, 0, msgs + offs[i], (uint32_t)(offs[i + 1] - offs[i]), dst.b, dst.len, lane2, no_clear);
  if (lane2 <= 0) store_g2_pt(out, i, h);
}

// =====================================================================================================
// point sums and multi-scalar multiplication.  Partials are RAW_PROJ structs in a workspace array.
// Stage 1: lane t accumulates items t, t + T, t + 2T, ... (optionally scaled by their scalars).
template <int G, int WITH_SCALARS>
__global__ void __launch_bounds__(BLS_BLOCK) k_accumulate(size_t n, const uint8_t* pts, int fmt, const uint8_t* scalars,
                                                        const uint32_t* perm, uint8_t* partials, size_t T) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  if (G == 1) {
    g1_jac acc, p;
    fp one;
    fp_one(one);
    jac_set_inf(acc);
    for (size_t i = t; i < n; i += T) {
      size_t src = perm ? perm[i] : i;
      load_g1_pt(p, pts, src, fmt);
      if (WITH_SCALARS) {
        uint32_t k[8];
        scalar_load_mod_r(k, scalars, i);
        jac_mul_scalar(p, p, k);
      }
      if (!WITH_SCALARS && fp_eq(p.z, one)) jac_madd(acc, acc, p.x, p.y);     // Z = 1 (deserialised, RAW_AFF): the mixed addition
      else jac_add(acc, acc, p);
    }
    store_g1_pt(partials, t, acc);
  } else {
    g2_jac acc, p;
    jac_set_inf(acc);
    for (size_t i = t; i < n; i += T) {
      size_t src = perm ? perm[i] : i;
      load_g2_pt(p, pts, src, fmt);
      if (WITH_SCALARS) {
        uint32_t k[8];
        scalar_load_mod_r(k, scalars, i);
        jac_mul_scalar(p, p, k);
      }
      jac_add(acc, acc, p);
    }
    store_g2_pt(partials, t, acc);
  }
}
// The same for G2 without scalars on two lanes per accumulator (MultiPublicKey::from_public_keys over G2 keys).  A key that
// comes from deserialisation has Z = 1 (so does every RAW_AFF input): it takes the mixed addition, 7M + 4S instead of 11M + 5S.
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_accumulate_g2s(size_t n, const uint8_t* pts, int fmt, uint8_t* partials, size_t T) {
  const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (t >= T) return;
  jac<hfp2> acc, p;
  hfp2 one, d;
  fe_one(one);
  jac_set_inf(acc);
  for (size_t i = t; i < n; i += T) {
    ld_g2s_fmt(p, pts, i, fmt);
    fe_sub(d, p.z, one);
    if (fe_is_zero(d)) jac_madd_body(acc, acc, p.x, p.y);     // (bodies, not the non-inlined functions: through those the accumulator
    else jac_add_body(acc, acc, p);                           //  travelled by reference, through scratch, once per key)
  }
  st_g2s(partials, t, acc);
}
// Stage 2 for G2 on two lanes per sum (jac<hfp2>): a fold level is one addition deep, i.e. latency-bound, and the
// lane-split addition is less than half as long as the one-lane one (0.10 -> 0.05 ms per level; folding the last levels
// inside one workgroup instead of one launch per level was measured too: no further gain)
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_point_fold_g2s(size_t m, size_t half, uint8_t* partials) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (i + half >= m || i >= half) return;
  jac<hfp2> a, b;
  ld_g2s(a, partials, i);
  ld_g2s(b, partials, i + half);
  jac_add_body(a, a, b);
  st_g2s(partials, i, a);
}
// Stage 2: partial[i] += partial[i + half]
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_point_fold(size_t m, size_t half, uint8_t* partials) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + half >= m || i >= half) return;
  if (G == 1) {
    g1_jac a, b;
    load_g1_pt(a, partials, i, 0);
    load_g1_pt(b, partials, i + half, 0);
    jac_add_body(a, a, b);
    store_g1_pt(partials, i, a);
  } else {
    g2_jac a, b;
    load_g2_pt(a, partials, i, 0);
    load_g2_pt(b, partials, i + half, 0);
    jac_add_body(a, a, b);
    store_g2_pt(partials, i, a);
  }
}

// grouped verification: the group's pair (sum_i r_i sig_i, -[c] g2) at slot c + 8 ng (the items' message points are uncleared:
// verify.cuh g2_negc_gen); an identity sum (every item excluded) is skipped
__global__ void __launch_bounds__(BLS_BLOCK) k_group_sigsum(size_t ng, size_t n, const uint8_t* scaled_sigs, uint32_t* pairs, int32_t* skip) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ng) return;
  g1_jac acc, p;
  jac_set_inf(acc);
  for (int k = 0; k < GROUPED_ITEMS; k++) {
    const size_t i = c * GROUPED_ITEMS + k;
    if (i >= n) break;
    load_g1_pt(p, scaled_sigs, i, 0);
    jac_add(acc, acc, p);
  }
  const size_t slot = c + (size_t)GROUPED_ITEMS * ng, stride = (GROUPED_ITEMS + 1) * ng;
  if (jac_is_inf(acc)) {
    skip[slot] = 1;
    return;
  }
  g1_aff S;
  g2_aff Q;
  jac_to_aff(S, acc);
  g2_negc_gen(Q);
  ws_st_pair(pairs, stride, slot, 0, S, Q);
  skip[slot] = 0;
}

// =====================================================================================================
// serialisation: any raw format -> compressed (modern or legacy) bytes
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_compress(size_t n, const uint8_t* pts, int fmt, int legacy, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (G == 1) {
    g1_jac p;
    g1_aff a;
    load_g1_pt(p, pts, i, fmt);
    jac_to_aff(a, p);
    uint8_t b[48];
    g1_compress(b, a, legacy != 0);
    for (int k = 0; k < 48; k++) out[i * 48 + k] = b[k];
  } else {
    g2_jac p;
    g2_aff a;
    load_g2_pt(p, pts, i, fmt);
    jac_to_aff(a, p);
    uint8_t b[96];
    g2_compress(b, a, legacy != 0);
    for (int k = 0; k < 96; k++) out[i * 96 + k] = b[k];
  }
}

template __global__ void k_accumulate<1, 0>(size_t, const uint8_t*, int, const uint8_t*, const uint32_t*, uint8_t*, size_t);
template __global__ void k_accumulate<1, 1>(size_t, const uint8_t*, int, const uint8_t*, const uint32_t*, uint8_t*, size_t);
template __global__ void k_accumulate<2, 0>(size_t, const uint8_t*, int, const uint8_t*, const uint32_t*, uint8_t*, size_t);
template __global__ void k_accumulate<2, 1>(size_t, const uint8_t*, int, const uint8_t*, const uint32_t*, uint8_t*, size_t);
template __global__ void k_point_fold<1>(size_t, size_t, uint8_t*);
template __global__ void k_point_fold<2>(size_t, size_t, uint8_t*);
template __global__ void k_compress<1>(size_t, const uint8_t*, int, int, uint8_t*);
template __global__ void k_compress<2>(size_t, const uint8_t*, int, int, uint8_t*);
#endif  // BLS_TU_POINTS

#if defined(BLS_TU_SIGN1) || defined(BLS_TU_SIGN2)
// =====================================================================================================
// sign side, used to build synthetic inputs (bench.py, tests): pk = sk * g, sig = sk * H(msg)
// (reference src/traits/sig_core.rs:108-117 core_sign, src/secret_key.rs:342-344 public_key).  sks: 32 B LE each.
template <int SG>
__global__ void __launch_bounds__(BLS_BLOCK) k_sign(size_t n, const uint8_t* sks, int aug, const uint8_t* msgs, const uint64_t* offs,
                                                  dst_arg dst, uint8_t* out_pks, uint8_t* out_sigs) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* sk = (const uint32_t*)(sks + 32 * i);
  const uint8_t* m = msgs + offs[i];
  uint32_t mlen = (uint32_t)(offs[i + 1] - offs[i]);
  if (SG == 1) {
    g2_aff g;
    fp2_load(g.x, G2_GEN_X);
    fp2_load(g.y, G2_GEN_Y);
    g.inf = false;
    g2_jac gj, pk;
    jac_from_aff(gj, g);
    jac_mul_scalar(pk, gj, sk);
    store_g2_pt(out_pks, i, pk);
    uint8_t pre[96];
    uint32_t pre_len = 0;
    if (aug) {
      g2_aff a;
      jac_to_aff(a, pk);
      g2_compress(pre, a, false);
      pre_len = 96;
    }
    g1_jac h;
    hash_to_g1(h, pre, pre_len, m, mlen, dst.b, dst.len);
    jac_mul_scalar(h, h, sk);
    store_g1_pt(out_sigs, i, h);
  } else {
    g1_aff g;
    fp_load(g.x, G1_GEN_X);
    fp_load(g.y, G1_GEN_Y);
    g.inf = false;
    g1_jac gj, pk;
    jac_from_aff(gj, g);
    jac_mul_scalar(pk, gj, sk);
    store_g1_pt(out_pks, i, pk);
    uint8_t pre[48];
    uint32_t pre_len = 0;
    if (aug) {
      g1_aff a;
      jac_to_aff(a, pk);
      g1_compress(pre, a, false);
      pre_len = 48;
    }
    g2_jac h;
    hash_to_g2(h, pre, pre_len, m, mlen, dst.b, dst.len);
    jac_mul_scalar(h, h, sk);
    store_g2_pt(out_sigs, i, h);
  }
}
// Signature proof of knowledge, BlsSignatureProof::verify (reference src/traits/sig_proof.rs:102-142): the checks in the
// reference's order (commitment, proof, pk identity; y zero), T = commitment + y * H(msg), and the pairs of
//   e(proof, g) * e(T, pk) == 1   written as   e(T, pk) * e(-proof, -g) == 1,
// i.e. core_verify's layout with H(m) := T and sig := -proof, so that the pairing stages (and the fixed -g2 lines) are shared.
template <int SG>
__global__ void __launch_bounds__(BLS_BLOCK) k_prepare_proof(size_t n, const uint8_t* commitments, const uint8_t* proofs,
                                                           const uint8_t* pks, const uint8_t* ys, int fmt, const uint8_t* msgs,
                                                           const uint64_t* offs, dst_arg dst, uint32_t* pairs, int32_t* status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* y = (const uint32_t*)(ys + 32 * i);
  const uint8_t* m = msgs + offs[i];
  uint32_t mlen = (uint32_t)(offs[i + 1] - offs[i]);
  g1_aff P[2];
  g2_aff Q[2];
  int st = BLS_OK;
  if (SG == 1) {
    g1_jac u, v, a;
    g2_jac pk;
    load_g1_pt(u, commitments, i, fmt);
    load_g1_pt(v, proofs, i, fmt);
    load_g2_pt(pk, pks, i, fmt);
    if (jac_is_inf(u)) st = BLS_ERR_COMMITMENT_IDENTITY;
    else if (jac_is_inf(v)) st = BLS_ERR_PROOF_IDENTITY;
    else if (jac_is_inf(pk)) st = BLS_ERR_PK_IDENTITY;
    else if (words_all_zero(y, 8)) st = BLS_ERR_ZERO_CHALLENGE;
    if (st == BLS_OK) {
      hash_to_g1(a, nullptr, 0, m, mlen, dst.b, dst.len);
      jac_mul_scalar(a, a, y);
      jac_add(a, a, u);
      if (jac_is_inf(a)) {          // e(T, pk) = 1: the product is e(proof, g) with proof != identity, never one
        st = BLS_ERR_INVALID_SIGNATURE;
      } else {
        jac_neg(v, v);
        g1g2_to_aff(P[1], Q[0], v, pk);
        jac_to_aff(P[0], a);
        g2_neg_gen(Q[1]);
      }
    }
  } else {
    g2_jac u, v, a;
    g1_jac pk;
    load_g2_pt(u, commitments, i, fmt);
    load_g2_pt(v, proofs, i, fmt);
    load_g1_pt(pk, pks, i, fmt);
    if (jac_is_inf(u)) st = BLS_ERR_COMMITMENT_IDENTITY;
    else if (jac_is_inf(v)) st = BLS_ERR_PROOF_IDENTITY;
    else if (jac_is_inf(pk)) st = BLS_ERR_PK_IDENTITY;
    else if (words_all_zero(y, 8)) st = BLS_ERR_ZERO_CHALLENGE;
    if (st == BLS_OK) {
      hash_to_g2(a, nullptr, 0, m, mlen, dst.b, dst.len);
      jac_mul_scalar(a, a, y);
      jac_add(a, a, u);
      if (jac_is_inf(a)) {
        st = BLS_ERR_INVALID_SIGNATURE;
      } else {
        jac_neg(v, v);
        g1g2_to_aff(P[0], Q[1], pk, v);
        jac_to_aff(Q[0], a);
        g1_neg_gen(P[1]);
      }
    }
  }
  status[i] = st;
  if (st != BLS_OK) return;
  ws_st_pair(pairs, n, i, 0, P[0], Q[0]);
  ws_st_pair(pairs, n, i, 1, P[1], Q[1]);
}
#if defined(BLS_TU_SIGN1)
template __global__ void k_sign<1>(size_t, const uint8_t*, int, const uint8_t*, const uint64_t*, dst_arg, uint8_t*, uint8_t*);
template __global__ void k_prepare_proof<1>(size_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, int, const uint8_t*, const uint64_t*, dst_arg, uint32_t*, int32_t*);
#else
template __global__ void k_sign<2>(size_t, const uint8_t*, int, const uint8_t*, const uint64_t*, dst_arg, uint8_t*, uint8_t*);
template __global__ void k_prepare_proof<2>(size_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, int, const uint8_t*, const uint64_t*, dst_arg, uint32_t*, int32_t*);
#endif
#endif  // BLS_TU_SIGN*

#if defined(BLS_TU_MILLERS) || defined(BLS_TU_FINALEXPS) || defined(BLS_TU_LINES) || defined(BLS_TU_MILLERF) || defined(BLS_TU_FINALEXP2) || defined(BLS_TU_CYCRUN4)
// =====================================================================================================
// lane-split variants (tower_split.cuh): two adjacent lanes per item, 64-thread workgroups = 32 items
#include "tower_split.cuh"
__device__ __forceinline__ void ws_ld_hfp2(hfp2& r, const uint32_t* ws, size_t stride, size_t i, int w0) {
  ws_ld_fp(r.v, ws, stride, i, w0 + (lane_hi() ? W1 : 0));
}
__device__ __forceinline__ void ws_st_hfp2(uint32_t* ws, size_t stride, size_t i, int w0, const hfp2& a) {
  ws_st_fp(ws, stride, i, w0 + (lane_hi() ? W1 : 0), a.v);
}
// Wave-uniform arguments of the non-inlined operation functions.  A function's arguments arrive in VECTOR registers and its pointers
// are generic: left alone, every row access of a workspace becomes a flat load through a 64-bit per-lane address (two 64-bit vector
// additions per row: 139 of them per call of the accumulator's line product, and their spill slots when a row is read twice).
// The workspace pointers, strides, entry and slot numbers ARE wave-uniform, so the functions say so: v_readfirstlane makes them
// scalars, the pointer becomes a global (address space 1) pointer, and an access is global_load saddr + this lane's 32-bit offset.
typedef __attribute__((address_space(1))) uint32_t g_u32;
typedef __attribute__((address_space(1))) const uint32_t gc_u32;
__device__ __forceinline__ uint32_t uni_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ size_t uni_sz(size_t v) { return ((size_t)uni_u32((uint32_t)(v >> 32)) << 32) | uni_u32((uint32_t)v); }
__device__ __forceinline__ g_u32* uni_global(uint32_t* p) { return (g_u32*)uni_sz((size_t)p); }
__device__ __forceinline__ gc_u32* uni_global(const uint32_t* p) { return (gc_u32*)uni_sz((size_t)p); }
// word `w` (a lane offset whose BYTE offset fits 32 bits: the hosts keep rows below 2^30 words) of a uniform row
__device__ __forceinline__ uint32_t g_ld(gc_u32* row, uint32_t w) { return *(gc_u32*)((__attribute__((address_space(1))) const char*)row + (w << 2)); }
__device__ __forceinline__ void g_st(g_u32* row, uint32_t w, uint32_t x) { *(g_u32*)((__attribute__((address_space(1))) char*)row + (w << 2)) = x; }
// ... for data that streams through once (a chunk's line values: 2.5 GB written by the line kernel, read once by the accumulator's): non-temporal,
// so that it does not push the spill slots of the functions in between out of the XCD's L2
__device__ __forceinline__ uint32_t g_ld_nt(gc_u32* row, uint32_t w) { return __builtin_nontemporal_load((gc_u32*)((__attribute__((address_space(1))) const char*)row + (w << 2))); }
__device__ __forceinline__ void g_st_nt(g_u32* row, uint32_t w, uint32_t x) { __builtin_nontemporal_store(x, (g_u32*)((__attribute__((address_space(1))) char*)row + (w << 2))); }
__device__ __forceinline__ void ws_ld_hfp12(fp12_t<hfp2>& f, const uint32_t* ws, size_t stride, size_t i) {
  ws_ld_hfp2(f.c0.a0, ws, stride, i, 0);
  ws_ld_hfp2(f.c0.a1, ws, stride, i, W2);
  ws_ld_hfp2(f.c0.a2, ws, stride, i, 2 * W2);
  ws_ld_hfp2(f.c1.a0, ws, stride, i, 3 * W2);
  ws_ld_hfp2(f.c1.a1, ws, stride, i, 4 * W2);
  ws_ld_hfp2(f.c1.a2, ws, stride, i, 5 * W2);
}
__device__ __forceinline__ void ws_st_hfp12(uint32_t* ws, size_t stride, size_t i, const fp12_t<hfp2>& f) {
  ws_st_hfp2(ws, stride, i, 0, f.c0.a0);
  ws_st_hfp2(ws, stride, i, W2, f.c0.a1);
  ws_st_hfp2(ws, stride, i, 2 * W2, f.c0.a2);
  ws_st_hfp2(ws, stride, i, 3 * W2, f.c1.a0);
  ws_st_hfp2(ws, stride, i, 4 * W2, f.c1.a1);
  ws_st_hfp2(ws, stride, i, 5 * W2, f.c1.a2);
}
#endif

#if defined(BLS_TU_MILLERS)
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) k_miller2s(size_t n, const uint32_t* pairs, const int32_t* status, uint32_t* fws, int fixed_g2) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (i >= n) return;
  if (status[i] != BLS_OK) return;
  g1_aff P[2];
  aff<hfp2> Q[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int w0 = k * 3 * W2;
    ws_ld_fp(P[k].x, pairs, n, i, w0);
    ws_ld_fp(P[k].y, pairs, n, i, w0 + W1);
    ws_ld_hfp2(Q[k].x, pairs, n, i, w0 + W2);
    ws_ld_hfp2(Q[k].y, pairs, n, i, w0 + 2 * W2);
    P[k].inf = false;
    Q[k].inf = false;
  }
  __shared__ uint32_t fsh[F12_SH_WORDS * BLS_BLOCK];   // the accumulator, packed (tower_split.cuh)
  f12_sh acc = {lds_column(fsh)};
  if (fixed_g2) miller_loop_fixed_g2(acc, P[0], Q[0], P[1], fixed_g2 == 2 ? G2NEGC_LINES : G2NEG_LINES);   // 2: pair 1 is (sig, -[c] g2)
  else miller_loop<2>(acc, P, Q);
  fp12_t<hfp2> f;
  sh_ld_f12(f, acc.sh);
  fp12_conj(f, f);
  ws_st_hfp12(fws, n, i, f);
}
#endif

#if defined(BLS_TU_MILLERS)
// Pairing products (aggregate verify / pairing_product_is_one): every lane pair runs ONE Miller loop over MILLER1_GROUP
// items (g, g + q, g + 2q, ...; q = ceil(n / MILLER1_GROUP)), so that the accumulator squarings are shared -- the product
// over all items is all that is needed.  Writes q partial products; skipped items contribute 1.
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) k_miller1s(size_t n, size_t stride, const uint32_t* pairs, const int32_t* skip, uint32_t* fws) {
  const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1, q = (n + MILLER1_GROUP - 1) / MILLER1_GROUP;
  if (g >= q) return;
  __shared__ uint32_t fsh[F12_SH_WORDS * BLS_BLOCK];
  f12_sh acc = {lds_column(fsh)};
  g1_aff P[MILLER1_GROUP];
  aff<hfp2> Q[MILLER1_GROUP];
#pragma unroll
  for (int k = 0; k < MILLER1_GROUP; k++) {
    const size_t j = g + (size_t)k * q;
    const bool have = j < n;
    const size_t jj = have ? j : g;
    ws_ld_fp(P[k].x, pairs, stride, jj, 0);
    ws_ld_fp(P[k].y, pairs, stride, jj, W1);
    ws_ld_hfp2(Q[k].x, pairs, stride, jj, W2);
    ws_ld_hfp2(Q[k].y, pairs, stride, jj, 2 * W2);
    P[k].inf = !have || skip[jj] != 0;
    Q[k].inf = false;
  }
  miller_loop<MILLER1_GROUP>(acc, P, Q);
  fp12_t<hfp2> f;
  sh_ld_f12(f, acc.sh);
  fp12_conj(f, f);
  ws_st_hfp12(fws, stride, g, f);
}
#endif

#if defined(BLS_TU_LINES) || defined(BLS_TU_MILLERF)
// =====================================================================================================
// Round 3: the two-pair Miller loop cut in two kernels so that neither holds more state than its registers and LDS take
// ("The Miller loop in two kernels").  k_lines2s walks the G2 point(s), evaluates the lines, merges the two
// line values of every step (tower.cuh lines_merge) and streams the five coefficients per step to HBM; k_millerf2s owns nothing
// but the accumulator f in LDS and multiplies the stream into it: one squaring and ONE semi-sparse product per step.  Both inline
// everything but the multiplier leaves (BLS_INLINE_MILLER = 1 in their translation units): no operand travels by reference.
// Line workspace: word w (0..69: c0, c2, c4, c3, c5, fourteen limbs each, this lane's component) of entry e of lane t at
// lines[((size_t)e * LINE5_WORDS + w) * lanes + t] -- every access a coalesced 256-byte wave access; 68 x 70 x 4 B = 19 KB per
// lane.  The kernels work on the items [first, first + count) of a batch of n (the host walks large batches in chunks).
// (the row pointer is wave-uniform and the lane offset 32 bits wide, so that every access is the SGPR-base + VGPR-offset form:
// with per-lane 64-bit addresses the compiler kept 140 address registers alive across the loop)
__device__ __forceinline__ void line5_st(uint32_t* lines, size_t lanes, uint32_t t, int e, const line5_t<hfp2>& L) {
  const fp* c[5] = {&L.c0.v, &L.c2.v, &L.c4.v, &L.c3.v, &L.c5.v};
  lanes = uni_sz(lanes);
  g_u32* row = uni_global(lines) + (size_t)uni_u32((uint32_t)e) * LINE5_WORDS * lanes;
#pragma unroll
  for (int j = 0; j < 5; j++)
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      g_st_nt(row, t, (uint32_t)c[j]->l[k]);
      row += lanes;
    }
}
__device__ __forceinline__ void line5_ld(line5_t<hfp2>& L, const uint32_t* lines, size_t lanes, uint32_t t, int e) {
  fp* c[5] = {&L.c0.v, &L.c2.v, &L.c4.v, &L.c3.v, &L.c5.v};
  lanes = uni_sz(lanes);
  gc_u32* row = uni_global(lines) + (size_t)uni_u32((uint32_t)e) * LINE5_WORDS * lanes;
#pragma unroll
  for (int j = 0; j < 5; j++)
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      c[j]->l[k] = (int32_t)g_ld_nt(row, t);
      row += lanes;
    }
}
#endif

#if defined(BLS_TU_LINES)
// State of the point walk in LDS, one column per lane (word k at sh[k * 64]), unpacked: T = (X, Y, Z) 3 x 14 words, then one Fp
// of P0 and one of P1 -- the even lane keeps the x coordinates, the odd lane the y coordinates, and the partner's word comes by
// DPP where the other one is needed.  70 of the 80 dwords a lane may use at two waves per SIMD.
#define LS_TX 0
#define LS_TY FP_NL
#define LS_TZ (2 * FP_NL)
#define LS_P0 (3 * FP_NL)
#define LS_P1 (4 * FP_NL)
#define LS_WORDS (5 * FP_NL)
__device__ __forceinline__ void ls_ld(fp& r, const lds_u32* sh, int w0) {
#pragma unroll
  for (int k = 0; k < FP_NL; k++) r.l[k] = (int32_t)sh[(w0 + k) * BLS_SH_STRIDE];
}
__device__ __forceinline__ void ls_st(lds_u32* sh, int w0, const fp& a) {
#pragma unroll
  for (int k = 0; k < FP_NL; k++) sh[(w0 + k) * BLS_SH_STRIDE] = (uint32_t)a.l[k];
}
// the x (want_y = false) or y coordinate of the point stored split over the lane pair at w0, on both lanes
__device__ __forceinline__ void ls_ld_coord(fp& r, const lds_u32* sh, int w0, bool want_y) {
  fp own, other;
  ls_ld(own, sh, w0);
  fp_partner(other, own);
  fp_sel(r, lane_hi() == want_y, own, other);
}
struct miller_lds {
  lds_u32* sh;
  const uint32_t* pairs;   // the pairs workspace: Q0 is read from it at the five addition steps
  size_t n, i;
  BLS_MFN void ld_tx(hfp2& r) const { ls_ld(r.v, sh, LS_TX); }
  BLS_MFN void ld_ty(hfp2& r) const { ls_ld(r.v, sh, LS_TY); }
  BLS_MFN void ld_tz(hfp2& r) const { ls_ld(r.v, sh, LS_TZ); }
  BLS_MFN void st_tx(const hfp2& v) const { ls_st(sh, LS_TX, v.v); }
  BLS_MFN void st_ty(const hfp2& v) const { ls_st(sh, LS_TY, v.v); }
  BLS_MFN void st_tz(const hfp2& v) const { ls_st(sh, LS_TZ, v.v); }
  BLS_MFN void ld_xp(fp& r) const { ls_ld_coord(r, sh, LS_P0, false); }
  BLS_MFN void ld_yp(fp& r) const { ls_ld_coord(r, sh, LS_P0, true); }
  BLS_MFN void ld_xq(hfp2& r) const { ws_ld_hfp2(r, pairs, n, i, W2); }
  BLS_MFN void ld_yq(hfp2& r) const { ws_ld_hfp2(r, pairs, n, i, 2 * W2); }
};
// One entry of the loop: walk this kernel's point (doubling or addition), evaluate the line at its G1 point, then
//   SRC_TABLE  merge with the line of a FIXED second G2 argument (normalised table row at P1) and store the five coefficients;
//   SRC_NONE   store the three line coefficients as they are (first pass over two general pairs: pair 0);
//   SRC_LINES  merge with the line the first pass stored for this entry (second pass: pair 1) and store the five coefficients.
// A function whose arguments are scalars; it keeps nothing in registers between calls (see k_millerf2s about disable_tail_calls).
#define SRC_TABLE 0
#define SRC_NONE 1
#define SRC_LINES 2
#define LINE3_WORDS (3 * FP_NL)
__device__ __forceinline__ void line3_st(uint32_t* lines3, size_t lanes, uint32_t t, int e, const hfp2& l0, const hfp2& l2, const hfp2& l3) {
  const fp* c[3] = {&l0.v, &l2.v, &l3.v};
  lanes = uni_sz(lanes);
  g_u32* row = uni_global(lines3) + (size_t)uni_u32((uint32_t)e) * LINE3_WORDS * lanes;
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      g_st_nt(row, t, (uint32_t)c[j]->l[k]);
      row += lanes;
    }
}
__device__ __forceinline__ void line3_ld(hfp2& l0, hfp2& l2, hfp2& l3, const uint32_t* lines3, size_t lanes, uint32_t t, int e) {
  fp* c[3] = {&l0.v, &l2.v, &l3.v};
  lanes = uni_sz(lanes);
  gc_u32* row = uni_global(lines3) + (size_t)uni_u32((uint32_t)e) * LINE3_WORDS * lanes;
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      c[j]->l[k] = (int32_t)g_ld_nt(row, t);
      row += lanes;
    }
}
template <int ADD, int SRC>
static __device__ __noinline__ void lines_step_fn(lds_u32* sh, const uint32_t* pairs, size_t n, size_t i, const uint32_t* row, uint32_t* lines, uint32_t* lines3,
                                                  size_t lanes, uint32_t t, int e, int skip_a = 0, int skip_b = 0) {
  const miller_lds st = {sh, pairs, n, i};
  hfp2 l0, l2, l3;
  if (ADD) miller_add_step_at(st, l0, l2, l3);
  else miller_dbl_step_at(st, l0, l2, l3);
  if (SRC == SRC_NONE) {
    line3_st(lines3, lanes, t, e, l0, l2, l3);
    return;
  }
  line5_t<hfp2> L;
  if (SRC == SRC_TABLE) {
    fp x1, y1;
    hfp2 n0, n2, c;
    {                                              // this lane's components of the table row (a wave-uniform pointer: see uni_global)
      gc_u32* tr = uni_global(row);
      const uint32_t o = lane_hi() ? FP_NL : 0;
#pragma unroll
      for (int k = 0; k < FP_NL; k++) {
        n0.v.l[k] = (int32_t)g_ld(tr, o + k);
        c.v.l[k] = (int32_t)g_ld(tr, o + 2 * FP_NL + k);
      }
    }
    ls_ld_coord(x1, sh, LS_P1, false);
    fp2_mul_fp(n2, c, x1);
    ls_ld_coord(y1, sh, LS_P1, true);
    lines_merge_y(L, l0, l2, l3, n0, n2, y1);
  } else {
    hfp2 m0, m2, m3;
    line3_ld(m0, m2, m3, lines3, lanes, t, e);
    lines_merge(L, m0, m2, m3, l0, l2, l3);        // pair 0's line first, as miller_line5_pair does
    if (skip_a | skip_b) {                         // pairing products: a pair that contributes 1 leaves the other's line, or 1
      hfp2 z, one;
      fp2_zero(z);
      fp2_one(one);
      const hfp2& s0 = skip_a ? l0 : m0;
      const hfp2& s2 = skip_a ? l2 : m2;
      const hfp2& s3 = skip_a ? l3 : m3;
      const bool both = skip_a && skip_b;
      fp2_reduce(L.c0, s0);
      fp2_reduce(L.c2, s2);
      fp2_reduce(L.c3, s3);
      fp2_cmov(L.c0, one, both);
      fp2_cmov(L.c2, z, both);
      fp2_cmov(L.c3, z, both);
      L.c4 = z;
      L.c5 = z;
    }
  }
  line5_st(lines, lanes, t, e, L);
}
// fixed_g2: 1 / 2 = the second pair's G2 member is -g2 / -[c] g2 (normalised tables of g2neg_lines.cuh): one launch, pass = 0.
// fixed_g2 = 0, two general pairs: two launches -- pass 1 walks pair 0 and stores its lines (lines3), pass 2 walks pair 1, merges
// and stores the five coefficients -- because two points do not fit the 80 LDS dwords a lane has at two waves per SIMD.
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_lines2s(size_t n, size_t first, size_t count, const uint32_t* pairs, const int32_t* status, uint32_t* lines, uint32_t* lines3, size_t lanes, int fixed_g2, int pass) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t >> 1;
  if (j >= count) return;
  const size_t i = first + j;
  if (status[i] != BLS_OK) return;
  __shared__ uint32_t lsh[LS_WORDS * BLS_BLOCK];
  lds_u32* sh = lds_column(lsh);
  const int pair = pass == 2 ? 1 : 0;                  // which (P, Q) of the workspace this launch walks
  const uint32_t* pw = pairs + (size_t)pair * 3 * W2 * n;   // the accessors read Q at word offsets W2 and 2 W2 of the pair they are given
  {
    hfp2 q;
    fp p;
    ws_ld_hfp2(q, pw, n, i, W2);
    ls_st(sh, LS_TX, q.v);
    ws_ld_hfp2(q, pw, n, i, 2 * W2);
    ls_st(sh, LS_TY, q.v);
    fp2_one(q);
    ls_st(sh, LS_TZ, q.v);
    ws_ld_fp(p, pw, n, i, lane_hi() ? W1 : 0);                   // this pair's G1 point: x on the even lane, y on the odd lane
    ls_st(sh, LS_P0, p);
    ws_ld_fp(p, pairs, n, i, 3 * W2 + (lane_hi() ? W1 : 0));     // P1 (used by the table form only)
    ls_st(sh, LS_P1, p);
  }
  if (pass == 0) {
    const uint32_t (*rows)[4 * FP_NL] = fixed_g2 == 2 ? G2NEGC_LINES_N : G2NEG_LINES_N;
    for (int e = 0; e < MILLER_ENTRIES; e++) {
      if (miller_entry_is_add(e)) lines_step_fn<1, SRC_TABLE>(sh, pw, n, i, rows[e], lines, lines3, lanes, t, e);
      else lines_step_fn<0, SRC_TABLE>(sh, pw, n, i, rows[e], lines, lines3, lanes, t, e);
    }
  } else if (pass == 1) {
    for (int e = 0; e < MILLER_ENTRIES; e++) {
      if (miller_entry_is_add(e)) lines_step_fn<1, SRC_NONE>(sh, pw, n, i, nullptr, lines, lines3, lanes, t, e);
      else lines_step_fn<0, SRC_NONE>(sh, pw, n, i, nullptr, lines, lines3, lanes, t, e);
    }
  } else {
    for (int e = 0; e < MILLER_ENTRIES; e++) {
      if (miller_entry_is_add(e)) lines_step_fn<1, SRC_LINES>(sh, pw, n, i, nullptr, lines, lines3, lanes, t, e);
      else lines_step_fn<0, SRC_LINES>(sh, pw, n, i, nullptr, lines, lines3, lanes, t, e);
    }
  }
}
#endif

#if defined(BLS_TU_LINES)
// Pairing PRODUCTS (aggregate verify: reference src/traits/sig_core.rs:149-178): mm one-pair items in a workspace of stride
// `stride`, flagged items contribute 1.  Two items share a merged line value: virtual item v pairs item v with item v + half
// (half = ceil(mm / 2)).  pass 1 walks item v and stores its lines, pass 2 walks item v + half, merges and stores the five
// coefficients (the two-pass form of k_lines2s); this launch covers the virtual items [first_v, first_v + count_v).
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_linesp(size_t mm, size_t half, size_t stride, const uint32_t* pairs, const int32_t* bad, uint32_t* lines, uint32_t* lines3, size_t lanes, size_t first_v,
         size_t count_v, int pass) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t >> 1;
  if (j >= count_v) return;
  const size_t v = first_v + j, ib = v + half;
  const bool have_b = ib < mm;
  const int skip_a = bad[v] != 0, skip_b = !have_b || bad[ib] != 0;
  if (pass == 1 && skip_a) return;                          // nobody reads a skipped item's lines
  const size_t i = pass == 1 || !have_b ? v : ib;           // (an absent partner: walk item v again, its result is discarded)
  __shared__ uint32_t lsh[LS_WORDS * BLS_BLOCK];
  lds_u32* sh = lds_column(lsh);
  {
    hfp2 q;
    fp p;
    ws_ld_hfp2(q, pairs, stride, i, W2);
    ls_st(sh, LS_TX, q.v);
    ws_ld_hfp2(q, pairs, stride, i, 2 * W2);
    ls_st(sh, LS_TY, q.v);
    fp2_one(q);
    ls_st(sh, LS_TZ, q.v);
    ws_ld_fp(p, pairs, stride, i, lane_hi() ? W1 : 0);
    ls_st(sh, LS_P0, p);
  }
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    if (pass == 1) {
      if (miller_entry_is_add(e)) lines_step_fn<1, SRC_NONE>(sh, pairs, stride, i, nullptr, lines, lines3, lanes, t, e);
      else lines_step_fn<0, SRC_NONE>(sh, pairs, stride, i, nullptr, lines, lines3, lanes, t, e);
    } else {
      if (miller_entry_is_add(e)) lines_step_fn<1, SRC_LINES>(sh, pairs, stride, i, nullptr, lines, lines3, lanes, t, e, skip_a, skip_b);
      else lines_step_fn<0, SRC_LINES>(sh, pairs, stride, i, nullptr, lines, lines3, lanes, t, e, skip_a, skip_b);
    }
  }
}
#endif

#if defined(BLS_TU_LINES)
// ---- k_linesp4: the plain line values of a pairing product's items (what k_linesp pass 1 stores) on FOUR lanes per item -- two lane
// pairs A (lanes 0, 1 of a quad) and B (lanes 2, 3) that share the doubling step: for shards of at most 32,768 items, where the
// two-lane kernel leaves one wave per SIMD issuing at half rate and its 68-step chain sets the time (2.0 ms for 32,768 items as for
// 65,536).  Lanes of a wave run ONE instruction stream, so the step is a sequence of slots in each of which both pairs do the same
// kind of product on operands of their own:
//     slot 1 (product)   A: X Y            B: Z Z
//     slot 2 (squaring)  A: Y^2 = B_       B: (Y + Z)^2
//     slot 3 (squaring)  A: X^2            B: E^2            (E = 12 (1 + u) Z^2 is B's after slot 1)
//       exchange (DPP quad_perm [2,3,0,1]):  A gets E, B gets B_
//     slot 4 (product)   A: X Y (B_ - 3E)  B: B_ H          (H = (Y + Z)^2 - B_ - Z^2)   -> X3 = 2 ., Z3 = 4 .
//     slot 5 (squaring)  B: (B_ + 3E)^2    (A: the same, unused)                         -> Y3 = . - 12 E^2 on B;  exchange: A gets Y3
//     slot 6 (by Fp)     A: -3 X^2 xP      B: H yP                                       -> the line's w^2 and w^3 coefficients
// i.e. six multiplier passes (2 x 588 + 4 x 392 multiply-adds per lane) on the critical path instead of eleven (3 x 588 + 8 x 392):
// the arithmetic of pairing.cuh miller_dbl_step_at operation for operation, so the stored values are the same field elements.
// A holds X, Y and B holds Y, Z in their LDS columns; the five addition steps exchange what is missing and run the whole step
// (miller_add_step_at) redundantly on both pairs.  Line workspace and lane numbering as k_linesp: item j at lanes 2 j, 2 j + 1.
#define L4_X 0
#define L4_Y FP_NL
#define L4_Z (2 * FP_NL)
#define L4_PX (3 * FP_NL)
#define L4_PY (4 * FP_NL)
#define L4_WORDS (5 * FP_NL)
__device__ __forceinline__ void fp_cross4(fp& r, const fp& a) {   // the other pair's lane of the same parity
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], 0x4E, 0xF, 0xF, true);
}
__device__ __forceinline__ void fp2_sel(hfp2& r, bool c, const hfp2& a, const hfp2& b) { fp_sel(r.v, c, a.v, b.v); }
struct miller_lds4 {
  lds_u32* sh;
  const uint32_t* pairs;
  size_t n, i;
  BLS_MFN void ld_tx(hfp2& r) const { ls_ld(r.v, sh, L4_X); }
  BLS_MFN void ld_ty(hfp2& r) const { ls_ld(r.v, sh, L4_Y); }
  BLS_MFN void ld_tz(hfp2& r) const { ls_ld(r.v, sh, L4_Z); }
  BLS_MFN void st_tx(const hfp2& v) const { ls_st(sh, L4_X, v.v); }
  BLS_MFN void st_ty(const hfp2& v) const { ls_st(sh, L4_Y, v.v); }
  BLS_MFN void st_tz(const hfp2& v) const { ls_st(sh, L4_Z, v.v); }
  BLS_MFN void ld_xp(fp& r) const { ls_ld(r, sh, L4_PX); }
  BLS_MFN void ld_yp(fp& r) const { ls_ld(r, sh, L4_PY); }
  BLS_MFN void ld_xq(hfp2& r) const { ws_ld_hfp2(r, pairs, n, i, W2); }
  BLS_MFN void ld_yq(hfp2& r) const { ws_ld_hfp2(r, pairs, n, i, 2 * W2); }
};
// A stores l0 and l2 (words 0..27 of the entry), B stores l3 (words 28..41): t2 = this lane's lane number in the line workspace
__device__ __forceinline__ void line3_st4(uint32_t* lines3, size_t lanes, uint32_t t2, int e, bool pr, const hfp2& l0, const hfp2& l2, const hfp2& l3) {
  lanes = uni_sz(lanes);
  g_u32* row = uni_global(lines3) + (size_t)uni_u32((uint32_t)e) * LINE3_WORDS * lanes;
  hfp2 first;
  fp2_sel(first, pr, l3, l0);
  const uint32_t w = t2 + (pr ? (uint32_t)(2 * FP_NL) * (uint32_t)lanes : 0u);
#pragma unroll
  for (int k = 0; k < FP_NL; k++) {
    g_st_nt(row, w, (uint32_t)first.v.l[k]);
    row += lanes;
  }
  if (!pr) {
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      g_st_nt(row, t2, (uint32_t)l2.v.l[k]);
      row += lanes;
    }
  }
}
static __device__ __noinline__ void lines4_dbl_fn(lds_u32* sh, uint32_t* lines3, size_t lanes, uint32_t t2, int e, bool pr) {
  hfp2 r0, r1, u, w, m1, s2, s3, E, g, f, b, h, l0, m4, s5, snd, rcv;
  const lds_u32* c0 = sh + (pr ? L4_Y : L4_X) * BLS_SH_STRIDE;     // A: X, Y    B: Y, Z
  ls_ld(r0.v, c0, 0);
  ls_ld(r1.v, c0, FP_NL);
  fp2_sel(u, pr, r1, r0);
  fp2_mul(m1, u, r1);              // slot 1   A: a = X Y        B: c = Z^2
  fp2_add(w, r0, r1);
  fp2_norm(w, w);
  fp2_sel(w, pr, w, r1);
  fp2_sqr(s2, w);                  // slot 2   A: B_ = Y^2       B: (Y + Z)^2
  fp2_mul_xi(E, m1);
  fp2_dbl(g, E);
  fp2_add(E, g, E);
  fp2_reduce(E, E);
  fp2_dbl(E, E);
  fp2_dbl(E, E);
  fp2_norm(E, E);                  // B: E = 3 b' Z^2 = 12 (1 + u) Z^2
  fp2_sel(w, pr, E, r0);
  fp2_sqr(s3, w);                  // slot 3   A: X^2            B: E^2
  fp2_sel(snd, pr, E, s2);
  fp_cross4(rcv.v, snd.v);
  fp2_sel(b, pr, rcv, s2);         // B_ on both pairs
  fp2_sel(E, pr, E, rcv);          // E on both pairs
  fp2_dbl(f, E);
  fp2_add(f, f, E);                // F = 3E
  fp2_sub(h, s2, b);
  fp2_sub(h, h, m1);
  fp2_norm(h, h);                  // B: H = 2 Y Z
  fp2_sub(l0, b, E);
  fp2_norm(l0, l0);                // the line's constant coefficient (A stores it)
  fp2_sub(g, b, f);
  fp2_norm(g, g);
  fp2_sel(u, pr, b, m1);
  fp2_sel(w, pr, h, g);
  fp2_mul(m4, u, w);               // slot 4   A: X Y (B_ - F)   B: B_ H
  fp2_dbl(g, m4);
  fp2_dbl(w, g);
  fp2_sel(g, pr, w, g);
  fp2_reduce(g, g);
  ls_st(sh, pr ? L4_Z : L4_X, g.v);   // A: X3 = 2 X Y (B_ - F)    B: Z3 = 4 B_ H
  fp2_add(g, b, f);
  fp2_norm(g, g);
  fp2_sqr(s5, g);                  // slot 5   (B_ + F)^2
  fp2_dbl(g, s3);
  fp2_add(g, g, s3);               // B: 3 E^2        A: 3 X^2
  hfp2 n3;
  fp2_neg(n3, g);                  // A: -3 X^2
  fp2_norm(g, g);
  fp2_dbl(g, g);
  fp2_dbl(g, g);                   // B: 12 E^2
  fp2_sub(g, s5, g);
  fp2_reduce(g, g);                // B: Y3
  fp_cross4(rcv.v, g.v);
  fp2_sel(g, pr, g, rcv);
  ls_st(sh, L4_Y, g.v);            // Y3 on both pairs
  fp k;
  ls_ld(k, sh, pr ? L4_PY : L4_PX);
  fp2_sel(u, pr, h, n3);
  hfp2 l23;
  fp2_mul_fp(l23, u, k);           // slot 6   A: l2 = -3 X^2 xP    B: l3 = H yP
  line3_st4(lines3, lanes, t2, e, pr, l0, l23, l23);
}
static __device__ __noinline__ void lines4_add_fn(lds_u32* sh, const uint32_t* pairs, size_t n, size_t i, uint32_t* lines3, size_t lanes, uint32_t t2, int e, bool pr) {
  {                                // A lacks Z, B lacks X: one exchange completes both columns
    fp snd, rcv;
    ls_ld(snd, sh, pr ? L4_Z : L4_X);
    fp_cross4(rcv, snd);
    ls_st(sh, pr ? L4_X : L4_Z, rcv);
  }
  const miller_lds4 st = {sh, pairs, n, i};
  hfp2 l0, l2, l3;
  miller_add_step_at(st, l0, l2, l3);   // both pairs, the same operands: X3, Y3, Z3 in both columns
  line3_st4(lines3, lanes, t2, e, pr, l0, l2, l3);
}
// The line values of ONE item whose G2 member is the fixed -g2 (fixed_g2 = 1) or -[c] g2 (2) -- the signature's pair of an
// aggregate verification of Bls12381G1Impl -- in k_linesp's format, as item 0 of a line workspace: no point walk, the normalised
// table row of every entry (g2neg_lines.cuh *_LINES_N) evaluated at the pair's G1 point: n0 + (n2 x1) w^2 + y1 w^3.  One lane pair.
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) k_lines_fixed(size_t stride, const uint32_t* pairs, const int32_t* bad, uint32_t* lines3, size_t lanes, int fixed_g2) {
  if (blockIdx.x != 0 || threadIdx.x >= 2) return;
  if (bad[0] != 0) return;
  const uint32_t hi = threadIdx.x & 1u;
  fp x1, y1;
  ws_ld_fp(x1, pairs, stride, 0, 0);
  ws_ld_fp(y1, pairs, stride, 0, W1);
  const uint32_t (*rows)[4 * FP_NL] = fixed_g2 == 2 ? G2NEGC_LINES_N : G2NEG_LINES_N;
  hfp2 l3;
  fp2_from_fp(l3, y1);
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    hfp2 n0, c, l2;
    fp2_load(n0, rows[e]);
    fp2_load(c, rows[e] + 2 * FP_NL);
    fp2_mul_fp(l2, c, x1);
    line3_st(lines3, lanes, hi, e, n0, l2, l3);
  }
}
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_linesp4(size_t cnt, size_t stride, const uint32_t* pairs, const int32_t* bad, uint32_t* lines3, size_t lanes) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t >> 2;
  if (j >= cnt) return;
  if (bad[j] != 0) return;                                  // (the four lanes of an item together)
  const bool pr = ((t >> 1) & 1u) != 0;
  const uint32_t t2 = (uint32_t)(2 * j) + (t & 1u);
  __shared__ uint32_t lsh[L4_WORDS * BLS_BLOCK];
  lds_u32* sh = lds_column(lsh);
  {
    hfp2 q;
    fp p;
    ws_ld_hfp2(q, pairs, stride, j, W2);
    ls_st(sh, L4_X, q.v);
    ws_ld_hfp2(q, pairs, stride, j, 2 * W2);
    ls_st(sh, L4_Y, q.v);
    fp2_one(q);
    ls_st(sh, L4_Z, q.v);
    ws_ld_fp(p, pairs, stride, j, 0);
    ls_st(sh, L4_PX, p);
    ws_ld_fp(p, pairs, stride, j, W1);
    ls_st(sh, L4_PY, p);
  }
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    if (miller_entry_is_add(e)) lines4_add_fn(sh, pairs, stride, j, lines3, lanes, t2, e, pr);
    else lines4_dbl_fn(sh, lines3, lanes, t2, e, pr);
  }
}
#endif

#if defined(BLS_TU_MILLERF)
// The two operations of the accumulator's kernel are FUNCTIONS whose arguments are scalars (LDS column, line pointer): each
// fits the register file on its own (no spill besides the callee-saved registers it borrows, 32-64 bytes of scratch per lane:
// L1/L2-resident), whereas inlined into one loop the compiler's allocation left 496 bytes of spill slots per lane -- 65 MB per
// launch, more than the L2s hold, and a quarter of the waves' time in s_waitcnt (profiles/r03_pmc_millerf_inlined.json).
static __device__ __noinline__ void f12_sh_sqr_fn(lds_u32* sh) {
  fp12_t<hfp2> a, r;
  sh_ld_f12(a, sh);
  fp12_sqr_body(r, a);
  sh_st_f12(sh, r);
}
// (measured and not adopted, round 4: the five coefficients fetched where they are needed -- three for the first Fp6 product, two for the
// sparse one, all five again for the Karatsuba sum -- instead of held: the same 46 spilled registers, 70 more loads, +2 %)
static __device__ __noinline__ void f12_sh_mul_line5_fn(lds_u32* sh, const uint32_t* lines, size_t lanes, uint32_t t, int e) {
  line5_t<hfp2> L;
  line5_ld(L, lines, lanes, t, e);
  f12_sh_mul_line5(sh, L);
}
// disable_tail_calls: a call the optimiser marks `tail` makes LLVM keep the callee-saved-register convention for its callee (TargetFrameLowering::
// isSafeForNoCSROpt), and each of the two functions would then save and restore all 112 callee-saved VGPRs on every call
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_millerf2s(size_t n, size_t first, size_t count, const int32_t* status, const uint32_t* lines, size_t lanes, uint32_t* fws) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t >> 1;
  if (j >= count) return;
  const size_t i = first + j;
  if (status[i] != BLS_OK) return;
  __shared__ uint32_t fsh[F12_SH_WORDS * BLS_BLOCK];   // the accumulator, packed (tower_split.cuh)
  f12_sh acc = {lds_column(fsh)};
  line5_t<hfp2> L;
  line5_ld(L, lines, lanes, t, 0);
  acc_set_line5(acc, L);
  for (int e = 1; e < MILLER_ENTRIES; e++) {
    if (!miller_entry_is_add(e)) f12_sh_sqr_fn(acc.sh);
    f12_sh_mul_line5_fn(acc.sh, lines, lanes, t, e);
  }
  fp12_t<hfp2> f;
  sh_ld_f12(f, acc.sh);
  fp12_conj(f, f);
  ws_st_hfp12(fws, n, i, f);
}
#endif

#if defined(BLS_TU_FINALEXPS)
// final exponentiation of item 0 of a workspace on one lane pair
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) k_finalexp_ones(const uint32_t* fws, size_t stride, int32_t* verdict) {
  if (blockIdx.x != 0 || threadIdx.x >= 2) return;
  fp12_t<hfp2> f;
  ws_ld_hfp12(f, fws, stride, 0);
  int st = pairing_verdict(f);
  if (!lane_hi()) *verdict = st;
}
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) k_finalexps(size_t n, const uint32_t* fws, int32_t* status) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (i >= n) return;
  if (status[i] != BLS_OK) return;
  fp12_t<hfp2> f;
  ws_ld_hfp12(f, fws, n, i);
  int st = pairing_verdict(f);
  if (!lane_hi()) status[i] = st;
}
#endif

#if defined(BLS_TU_MILLERF)
// ... and the accumulator's kernel of a pairing product: lane pair g multiplies the merged line values of the virtual items
// g, g + q, g + 2q, ... (group of them) into ONE accumulator -- one squaring per entry for all of them -- and leaves partial
// product g at item `out0 + g` of the Fp12 workspace (stride as given).  lines: the chunk's line workspace, virtual item j of
// the chunk at lanes 2 j, 2 j + 1.
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_millerfp(size_t count_v, size_t q, int group, const uint32_t* lines, size_t lanes, uint32_t* fws, size_t stride, size_t out0) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t g = t >> 1;
  if (g >= q) return;
  const uint32_t hi = t & 1u;
  __shared__ uint32_t fsh[F12_SH_WORDS * BLS_BLOCK];
  f12_sh acc = {lds_column(fsh)};
  {
    line5_t<hfp2> L;
    line5_ld(L, lines, lanes, (uint32_t)(2 * g) + hi, 0);
    acc_set_line5(acc, L);
  }
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    if (e > 0 && !miller_entry_is_add(e)) f12_sh_sqr_fn(acc.sh);
    for (int k = e == 0 ? 1 : 0; k < group; k++) {
      const size_t v = g + (size_t)k * q;
      if (v < count_v) f12_sh_mul_line5_fn(acc.sh, lines, lanes, (uint32_t)(2 * v) + hi, e);
    }
  }
  fp12_t<hfp2> f;
  sh_ld_f12(f, acc.sh);
  fp12_conj(f, f);
  ws_st_hfp12(fws, stride, out0 + g, f);
}
#endif

#if defined(BLS_TU_MILLERF)
// ... and for shards of at most one machine round of lane pairs (65,536 items): ONE item per accumulator slot and its plain line
// values (pass 1 of k_linesp only; the 13-product sparse multiplication), `group` items per accumulator.  Below a machine round
// every kernel costs a whole round whatever its size (a wave's time does not depend on how many SIMDs are busy), so the merge --
// one more pass of line kernels for fewer products here -- does not pay: 11 against 14-16 ms for 32,768-65,536 pairs.
static __device__ __noinline__ void f12_sh_mul_line3_fn(lds_u32* sh, const uint32_t* lines3, size_t lanes, uint32_t t, int e) {
  hfp2 l0, l2, l3;
  fp* c[3] = {&l0.v, &l2.v, &l3.v};
  lanes = uni_sz(lanes);
  gc_u32* row = uni_global(lines3) + (size_t)uni_u32((uint32_t)e) * (3 * FP_NL) * lanes;
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      c[j]->l[k] = (int32_t)g_ld_nt(row, t);
      row += lanes;
    }
  f12_sh_mul_line3(sh, l0, l2, l3);
}
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_millerfp3(size_t count, size_t q, int group, const int32_t* bad, const uint32_t* lines3, size_t lanes, uint32_t* fws, size_t stride, size_t out0) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t g = t >> 1;
  if (g >= q) return;
  const uint32_t hi = t & 1u;
  __shared__ uint32_t fsh[F12_SH_WORDS * BLS_BLOCK];
  f12_sh acc = {lds_column(fsh)};
  acc_one(acc);
  for (int e = 0; e < MILLER_ENTRIES; e++) {
    if (e > 0 && !miller_entry_is_add(e)) f12_sh_sqr_fn(acc.sh);
    for (int k = 0; k < group; k++) {
      const size_t v = g + (size_t)k * q;
      if (v < count && bad[v] == 0) f12_sh_mul_line3_fn(acc.sh, lines3, lanes, (uint32_t)(2 * v) + hi, e);
    }
  }
  fp12_t<hfp2> f;
  sh_ld_f12(f, acc.sh);
  fp12_conj(f, f);
  ws_st_hfp12(fws, stride, out0 + g, f);
}
#endif

#if defined(BLS_TU_MILLERF)
// ---- pairing products, second form (round 3): the product over the ITEMS first, entry by entry, then ONE Horner chain ----------
// The Miller function of item i is f_i = (..((l_i0)^2 l_i1)^2 ..) over the 68 entries (no squaring before the five addition entries),
// and squaring is multiplicative: prod_i f_i = (..((P_0)^2 P_1)^2 ..) with P_e = prod_i l_ie.  So a pairing product needs 68
// independent products over the items' sparse line values -- short independent tasks, any number of them in flight, no
// 68-step accumulator chain per lane pair -- and 63 squarings IN ALL (the engine's Horner program, k_f12_horner_wide) instead of 63
// per accumulator.  Per line value: 3 (merge of two lines) + 4.25 (two merged values -> a full Fp12 value) + 4.5 (the fold tree)
// = 11.75 Fp2 multiplications, against 13 + 12 for one item per accumulator (k_millerfp3).
// k_line_quad: grid (q / 32, 68).  Lane pair j of entry e = blockIdx.y takes the line values (k_linesp pass 1) of the chunk's items
// j, j + q, j + 2 q, j + 3 q (absent or flagged ones count as the line 1), merges them two by two and leaves the product of the
// two merged values at position e * rout + oout + j of the Fp12 workspace fout (stride sout).
static __device__ __noinline__ void f12_sh_st_ws_fn(lds_u32* sh, uint32_t* ws, size_t stride, uint32_t off);
__device__ __forceinline__ void quad_line_ld(hfp2& l0, hfp2& l2, hfp2& l3, size_t count, const int32_t* bad, gc_u32* row, size_t lanes, size_t x, uint32_t hi) {
  if (x < count && bad[x] == 0) {
    fp* c[3] = {&l0.v, &l2.v, &l3.v};
    const uint32_t t = (uint32_t)(2 * x) + hi;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < FP_NL; k++) {
        c[j]->l[k] = (int32_t)g_ld_nt(row, t);
        row += lanes;
      }
  } else {
    fp2_one(l0);
    fp2_zero(l2);
    fp2_zero(l3);
  }
}
template <int FIRST>
static __device__ __noinline__ void quad_merge_fn(lds_u32* sh, size_t count, const int32_t* bad, const uint32_t* row_, size_t lanes, size_t xa, size_t xb, uint32_t hi) {
  hfp2 a0, a2, a3, b0, b2, b3;
  gc_u32* row = uni_global(row_);
  lanes = uni_sz(lanes);
  quad_line_ld(a0, a2, a3, count, bad, row, lanes, xa, hi);
  quad_line_ld(b0, b2, b3, count, bad, row, lanes, xb, hi);
  line5_t<hfp2> L;
  lines_merge(L, a0, a2, a3, b0, b2, b3);
  if (FIRST) {
    f12_sh acc = {sh};
    acc_set_line5(acc, L);
  } else {
    f12_sh_mul_line5(sh, L);
  }
}
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_line_quad(size_t count, size_t q, const int32_t* bad, const uint32_t* lines3, size_t lanes, uint32_t* fout, size_t sout, size_t rout, size_t oout) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t >> 1;
  if (j >= q) return;
  const uint32_t hi = t & 1u;
  const int e = (int)blockIdx.y;
  __shared__ uint32_t fsh[F12_SH_WORDS * BLS_BLOCK];
  lds_u32* sh = lds_column(fsh);
  const uint32_t* row = lines3 + (size_t)e * (3 * FP_NL) * lanes;
  quad_merge_fn<1>(sh, count, bad, row, lanes, j, j + q, hi);
  quad_merge_fn<0>(sh, count, bad, row, lanes, j + 2 * q, j + 3 * q, hi);
  f12_sh_st_ws_fn(sh, fout + (size_t)e * rout + oout, sout, (uint32_t)j + hi * (uint32_t)(W1 * sout));
}
// k_f12_fold4: one level of the 68 fold trees.  grid (qout / 32, 68): lane pair j of entry e multiplies the values at positions
// e * rin + j + k qout (k < fan -- four, or five where that saves a level --, j + k qout < qin) of fin and leaves the product at
// position e * rout + oout + j of fout.  fin and fout must not overlap.
// (the factor streams in from the workspace half by half, its first half twice: with all of it held beside the three Fp6 products
// the function spilled 700 bytes per lane; and every access is wave-uniform row pointer + 32-bit lane offset, see line5_st)
// wsu_*: `ws` points at word 0 of a wave-uniform position of an Fp12 workspace, `off` = this lane's item offset from there plus
// W1 * stride on the odd lane (the imaginary parts' rows); the host keeps W1 * stride + items below 2^30 words
__device__ __forceinline__ void wsu_ld_hfp6(fp6_t<hfp2>& r, const uint32_t* ws, size_t stride, uint32_t off, int half) {
  fp* c[3] = {&r.a0.v, &r.a1.v, &r.a2.v};
  stride = uni_sz(stride);
  gc_u32* row = uni_global(ws) + (size_t)(3 * W2 * half) * stride;
#pragma unroll
  for (int j = 0; j < 3; j++) {
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      c[j]->l[k] = (int32_t)g_ld_nt(row, off);
      row += stride;
    }
    row += (size_t)(W2 - FP_NL) * stride;
  }
}
__device__ __forceinline__ void wsu_st_hfp6(uint32_t* ws, size_t stride, uint32_t off, int half, const fp6_t<hfp2>& a) {
  const fp* c[3] = {&a.a0.v, &a.a1.v, &a.a2.v};
  stride = uni_sz(stride);
  g_u32* row = uni_global(ws) + (size_t)(3 * W2 * half) * stride;
#pragma unroll
  for (int j = 0; j < 3; j++) {
#pragma unroll
    for (int k = 0; k < FP_NL; k++) {
      g_st_nt(row, off, (uint32_t)c[j]->l[k]);
      row += stride;
    }
    row += (size_t)(W2 - FP_NL) * stride;
  }
}
static __device__ __noinline__ void f12_sh_mul_ws_fn(lds_u32* sh, const uint32_t* ws, size_t stride, uint32_t off) {
  fp6_t<hfp2> x, y, t0, t1, m;
  sh_ld_f6(x, sh, 0);
  wsu_ld_hfp6(y, ws, stride, off, 0);
  fp6_mul(t0, x, y);                    // a0 b0
  __builtin_amdgcn_sched_barrier(0);
  sh_ld_f6(m, sh, 39);
  fp6_add(x, x, m);
  fp6_reduce(x, x);
  sh_st_f6(sh, 0, x);                   // slot 0 <- a0 + a1 (a0 is used up)
  wsu_ld_hfp6(y, ws, stride, off, 1);
  fp6_mul(t1, m, y);                    // a1 b1
  __builtin_amdgcn_sched_barrier(0);
  fp6_mul_v(m, t1);
  fp6_add(m, t0, m);
  fp6_reduce(m, m);
  sh_st_f6(sh, 39, m);                  // slot 1 <- c0 = a0 b0 + v a1 b1, parked (a1 is used up)
  fp6_add(t0, t0, t1);
  fp6_norm(t0, t0);
  __builtin_amdgcn_sched_barrier(0);
  sh_ld_f6(x, sh, 0);
  wsu_ld_hfp6(t1, ws, stride, off, 0);
  fp6_add(y, y, t1);
  fp6_norm(y, y);
  fp6_mul(m, x, y);
  fp6_sub(m, m, t0);
  fp6_reduce(m, m);                     // c1 = (a0 + a1)(b0 + b1) - a0 b0 - a1 b1
  sh_ld_f6(x, sh, 39);
  sh_st_f6(sh, 0, x);
  sh_st_f6(sh, 39, m);
}
static __device__ __noinline__ void f12_sh_ld_ws_fn(lds_u32* sh, const uint32_t* ws, size_t stride, uint32_t off) {
  fp6_t<hfp2> x;
  wsu_ld_hfp6(x, ws, stride, off, 0);      // stored from the packed form: reduced
  sh_st_f6(sh, 0, x);
  wsu_ld_hfp6(x, ws, stride, off, 1);
  sh_st_f6(sh, 39, x);
}
static __device__ __noinline__ void f12_sh_st_ws_fn(lds_u32* sh, uint32_t* ws, size_t stride, uint32_t off) {
  fp6_t<hfp2> x;
  sh_ld_f6(x, sh, 0);
  wsu_st_hfp6(ws, stride, off, 0, x);
  sh_ld_f6(x, sh, 39);
  wsu_st_hfp6(ws, stride, off, 1, x);
}
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_f12_fold4(size_t qin, size_t qout, int fan, const uint32_t* fin, size_t sin, size_t rin, uint32_t* fout, size_t sout, size_t rout, size_t oout) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const uint32_t j = t >> 1;
  if (j >= qout) return;
  const uint32_t hi = t & 1u;
  const uint32_t* in = fin + (size_t)blockIdx.y * rin;
  const uint32_t off_in = j + hi * (uint32_t)(W1 * sin);
  __shared__ uint32_t fsh[F12_SH_WORDS * BLS_BLOCK];
  lds_u32* sh = lds_column(fsh);
  f12_sh_ld_ws_fn(sh, in, sin, off_in);
  for (int k = 1; k < fan; k++) {
    const size_t x = j + (size_t)k * qout;
    if (x < qin) f12_sh_mul_ws_fn(sh, in + (size_t)k * qout, sin, off_in);
  }
  f12_sh_st_ws_fn(sh, fout + (size_t)blockIdx.y * rout + oout, sout, j + hi * (uint32_t)(W1 * sout));
}
#endif

#if defined(BLS_TU_FINALEXP2)
// =====================================================================================================
// Round 3: the final exponentiation of the lane-split batch path as a sequence of operations on ONE accumulator A in LDS
// (tower_split.cuh layout: 78 packed words per lane) and a per-lane VALUE STORE in HBM for everything else -- the inputs, the
// few Fp12 values the exponentiation keeps, the six compressed powers of every a^x and their denominators' prefix products.
// Every operation is a function whose arguments are scalars (LDS column, store pointer, slot numbers): nothing travels by
// reference through scratch, nothing stays in registers between operations, and the kernel is marked disable_tail_calls so that
// no operation saves callee-saved registers (see k_millerf2s).  The arithmetic -- and therefore the bound proof of tests/hostsim --
// is that of pairing.cuh final_exponentiation / tower_split.cuh fp12_pow_x, operation for operation.
// Store layout: word w (0..83: six Fp in tower order c0.a0 c0.a1 c0.a2 c1.a0 c1.a1 c1.a2, this lane's component, fourteen limbs
// each) of slot s of lane t at vs[((size_t)s * VS_WORDS + w) * lanes + t]: coalesced 256-byte wave accesses.
#define VS_WORDS (6 * FP_NL)
#define VS_SLOTS 10      // 0: t (inverse, then the running value of the hard part)  1: f (after the easy part)  2: a of the current a^x
                         // 3..6: the six compressed powers (4 Fp each: two per slot .. packed as 6 x 4 = 24 Fp = 4 slots)  7: prefix products
struct vs_ref {
  uint32_t* p;
  size_t lanes;
  uint32_t t;
};
__device__ __forceinline__ void vs_ld(fp& r, const vs_ref& v, int slot, int j) {
  const size_t lanes = uni_sz(v.lanes);
  gc_u32* row = uni_global((const uint32_t*)v.p) + ((size_t)uni_u32((uint32_t)slot) * VS_WORDS + (size_t)j * FP_NL) * lanes;
#pragma unroll
  for (int k = 0; k < FP_NL; k++) {
    r.l[k] = (int32_t)g_ld(row, v.t);
    row += lanes;
  }
}
__device__ __forceinline__ void vs_st(const vs_ref& v, int slot, int j, const fp& a) {
  const size_t lanes = uni_sz(v.lanes);
  g_u32* row = uni_global(v.p) + ((size_t)uni_u32((uint32_t)slot) * VS_WORDS + (size_t)j * FP_NL) * lanes;
#pragma unroll
  for (int k = 0; k < FP_NL; k++) {
    g_st(row, v.t, (uint32_t)a.l[k]);
    row += lanes;
  }
}
__device__ __forceinline__ hfp2& f12_coef(fp12_t<hfp2>& f, int j) {
  return j == 0 ? f.c0.a0 : j == 1 ? f.c0.a1 : j == 2 ? f.c0.a2 : j == 3 ? f.c1.a0 : j == 4 ? f.c1.a1 : f.c1.a2;
}
// operand transforms of fx_load / fx_mul
#define FX_PLAIN 0
#define FX_CONJ 1
#define FX_FROB1 2
#define FX_FROB2 3
__device__ __forceinline__ void fx_fetch(fp12_t<hfp2>& b, const vs_ref& v, int slot, int mode) {
  vs_ld(b.c0.a0.v, v, slot, 0);
  vs_ld(b.c0.a1.v, v, slot, 1);
  vs_ld(b.c0.a2.v, v, slot, 2);
  vs_ld(b.c1.a0.v, v, slot, 3);
  vs_ld(b.c1.a1.v, v, slot, 4);
  vs_ld(b.c1.a2.v, v, slot, 5);
  if (mode == FX_CONJ) {
    fp12_conj(b, b);
  } else if (mode == FX_FROB1) {
    fp12_t<hfp2> r;
    fp12_frob<1>(r, b);
    b = r;
  } else if (mode == FX_FROB2) {
    fp12_t<hfp2> r;
    fp12_frob<2>(r, b);
    b = r;
  }
}
// A <- op(V[slot])
static __device__ __noinline__ void fx_load(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t, int slot, int mode) {
  const vs_ref v = {vp, lanes, t};
  fp12_t<hfp2> b;
  fx_fetch(b, v, slot, mode);
  fp12_reduce(b, b);        // only reduced elements pack
  sh_st_f12(sh, b);
}
// V[slot] <- A
static __device__ __noinline__ void fx_store(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t, int slot) {
  const vs_ref v = {vp, lanes, t};
#pragma unroll
  for (int j = 0; j < 6; j++) {
    fp x;
    sh_ld_fp(x, sh, 13 * j);
    vs_st(v, slot, j, x);
  }
}
// A <- A * op(V[slot])
static __device__ __noinline__ void fx_mul(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t, int slot, int mode) {
  const vs_ref v = {vp, lanes, t};
  fp12_t<hfp2> b;
  fx_fetch(b, v, slot, mode);
  f12_sh_mul_body(sh, b);
}
// A <- A^(p^2)
static __device__ __noinline__ void fx_frob2(lds_u32* sh) {
  fp12_t<hfp2> a, r;
  sh_ld_f12(a, sh);
  fp12_frob<2>(r, a);
  fp12_reduce(r, r);
  sh_st_f12(sh, r);
}
// A <- 1 / A
static __device__ __noinline__ void fx_inv(lds_u32* sh) {
  fp12_t<hfp2> a, r;
  sh_ld_f12(a, sh);
  fp12_inv_body(r, a);
  sh_st_f12(sh, r);         // fp12_inv returns reduced coefficients
}
// A <- A^2 for A in the cyclotomic subgroup (Granger-Scott, streamed through LDS)
static __device__ __noinline__ void fx_cyc_sqr(lds_u32* sh) { f12_sh_cyclotomic_sqr_body(sh); }
// The compressed part of a^|x| (tower_split.cuh fp12_pow_x): A holds a; 63 compressed squarings of (z2, z3, z4, z5) in their LDS
// slots, the six powers whose bits are set in |x| saved to the store (power k: Fp j of slot 3 + (4 k + j) / 6 ...).
__device__ __forceinline__ void cpow_slot(int k, int j, int& slot, int& idx) {   // Fp j (0..3: z2 z3 z4 z5) of saved power k (0..5)
  const int q = 4 * k + j;
  slot = 3 + q / 6;
  idx = q % 6;
}
#ifndef BLS_CYC_KARA
#define BLS_CYC_KARA 1       // 1: one lane per Fp4 squaring, one-lane Karatsuba products (tower_split.cuh f12_sh_cyc_c_sqr_kara_body); 0: round 3's lane-split squarings
#endif
#if BLS_CYC_KARA
static __device__ __noinline__ void fx_pow_run(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t) {
  const vs_ref v = {vp, lanes, t};
  {                                        // the four coordinates out of A's packed slots into the loop's layout (A is dead from here:
    fp z2, z3, z4, z5;                     // fx_pow_finish rebuilds it from the saved powers, the fallback from V[2])
    sh_ld_fp(z2, sh, 39);
    sh_ld_fp(z3, sh, 26);
    sh_ld_fp(z4, sh, 13);
    sh_ld_fp(z5, sh, 65);
    cyck_from_split(sh, z2, z3, z4, z5);
  }
  const bool hi = lane_hi();
  const lds_u32* pa = sh + (hi ? CYCK_Q : CYCK_P) * BLS_SH_STRIDE;     // the squared element a + b s: (P, Q) on the even lane, (Q, P) on the odd one
  const lds_u32* pb = sh + (hi ? CYCK_P : CYCK_Q) * BLS_SH_STRIDE;
  int k = 0;
  for (int i = 1; i <= BLS_XK_LAST; i++) {          // 61 squarings, four powers saved: the four-power chain of pairing.cuh fp12_pow_x_compressed4
    f12_sh_cyc_c_sqr_kara_body(sh, pa, pb);
    if ((BLS_XK_SAVE >> i) & 1) {
      fp x[4];
      cyck_to_split(x[0], x[1], x[2], x[3], sh);
#pragma unroll
      for (int j = 0; j < 4; j++) {        // z2, z3, z4, z5
        int slot, idx;
        cpow_slot(k, j, slot, idx);
        vs_st(v, slot, idx, x[j]);
      }
      k++;
    }
  }
}
#else
static __device__ __noinline__ void fx_pow_run(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t) {
  const vs_ref v = {vp, lanes, t};
  {                                        // the four coordinates out of A's packed slots into the unpacked layout of the loop (A is dead from here:
    fp z2, z3, z4, z5;                     // fx_pow_finish rebuilds it from the saved powers, the fallback from V[2])
    sh_ld_fp(z2, sh, 39);
    sh_ld_fp(z3, sh, 26);
    sh_ld_fp(z4, sh, 13);
    sh_ld_fp(z5, sh, 65);
    shu_st_fp(sh, CYCU_Z2, z2);
    shu_st_fp(sh, CYCU_Z3, z3);
    shu_st_fp(sh, CYCU_Z4, z4);
    shu_st_fp(sh, CYCU_Z5, z5);
  }
  int k = 0;
  for (int i = 1; i <= 63; i++) {
    f12_sh_cyc_c_sqr_unpacked_body(sh);
    if ((BLS_X_ABS >> i) & 1) {
#pragma unroll
      for (int j = 0; j < 4; j++) {        // z2, z3, z4, z5
        fp x;
        int slot, idx;
        shu_ld_fp(x, sh, j * FP_NL);
        cpow_slot(k, j, slot, idx);
        vs_st(v, slot, idx, x);
      }
      k++;
    }
  }
}
#endif
// ... and the rest: the six powers decompressed with one shared inversion (Montgomery's trick over the denominators 4 z2) and
// multiplied into A.  Returns false on the lanes of an item one of whose z2 vanishes (never seen for honest inputs): the caller
// then runs the plain chain for that item.
// NP = 6: the powers of |x|'s six set bits, multiplied together (i from NP - 1 down to 0 in one call).
// NP = 4 (the run above: the powers 2^16, 2^48, 2^57, 2^61 and the plain tail of pairing.cuh fp12_pow_x_compressed4) in TWO calls around the
// tail, so that no call sits inside the loop that carries the inversion's state (with fx_store / fx_cyc_sqr / fx_mul inside it the frame
// grew from 720 to 1,104 bytes): PART 0 takes i = 3, 2 -- A = a^(2^61) conj(a^(2^57)) = c -- and leaves the running inverse in the
// store (slot 7, position 5); the caller parks c in slot 6, squares three times and multiplies by conj(c); PART 1 takes i = 1, 0.
static __device__ __noinline__ void fx_mul(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t, int slot, int mode);
static __device__ __noinline__ void fx_store(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t, int slot);
static __device__ __noinline__ void fx_cyc_sqr(lds_u32* sh);
template <int NP, int PART>
static __device__ __noinline__ bool fx_pow_finish_n(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t) {
  const vs_ref v = {vp, lanes, t};
  hfp2 pre, d, inv, tt;
  if (PART == 0) {
    bool ok = true;
    for (int i = 0; i < NP; i++) {           // prefix products of the denominators -> slot 7
      int slot, idx;
      hfp2 z2;
      cpow_slot(i, 0, slot, idx);
      vs_ld(z2.v, v, slot, idx);
      if (fp2_is_zero(z2)) ok = false;
      fp2_dbl(tt, z2);
      fp2_dbl(tt, tt);
      fp2_norm(d, tt);                       // 4 z2
      if (i == 0) pre = d;
      else fp2_mul(pre, pre, d);
      vs_st(v, 7, i, pre.v);
    }
    if (!ok) return false;
    fp2_inv(inv, pre);
  } else {
    vs_ld(inv.v, v, 7, 5);
  }
  const int hi = PART == 0 ? NP - 1 : 1, lo = (NP == 4 && PART == 0) ? 2 : 0;
#pragma nounroll
  for (int i = hi; i >= lo; i--) {
    hfp2 di;
    cyc_c<hfp2> s;
    int slot, idx;
    cpow_slot(i, 0, slot, idx);
    vs_ld(s.z2.v, v, slot, idx);
    if (i) {
      hfp2 pm;
      vs_ld(pm.v, v, 7, i - 1);
      fp2_mul(di, inv, pm);                // 1 / d_i
      fp2_dbl(tt, s.z2);
      fp2_dbl(tt, tt);
      fp2_norm(d, tt);
      fp2_mul(inv, inv, d);
    } else {
      di = inv;
    }
    cpow_slot(i, 1, slot, idx);
    vs_ld(s.z3.v, v, slot, idx);
    cpow_slot(i, 2, slot, idx);
    vs_ld(s.z4.v, v, slot, idx);
    cpow_slot(i, 3, slot, idx);
    vs_ld(s.z5.v, v, slot, idx);
    fp12_t<hfp2> e;
    cyc_decompress(e, s, di);
    if (NP == 4) {                         // the second power enters as its conjugate (a sign, not a branch: the loop body stays one)
      const int32_t m = i == 2 ? -1 : 0;
      fp* hi3[3] = {&e.c1.a0.v, &e.c1.a1.v, &e.c1.a2.v};
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < FP_NL; k++) hi3[j]->l[k] = (hi3[j]->l[k] ^ m) - m;
    }
    if (i == NP - 1) {
      fp12_reduce(e, e);
      sh_st_f12(sh, e);
    } else {
      f12_sh_mul_body(sh, e);
    }
  }
  if (NP == 4 && PART == 0) {
    fp2_reduce(inv, inv);
    vs_st(v, 7, 5, inv.v);
  }
  return true;
}
static __device__ __noinline__ bool fx_pow_finish(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t) { return fx_pow_finish_n<6, 0>(sh, vp, lanes, t); }
// the plain chain on A (Granger-Scott squarings, five multiplications by a = V[2]): the fallback of fx_pow_finish
static __device__ __noinline__ void fx_pow_plain(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t) {
  fx_load(sh, vp, lanes, t, 2, FX_PLAIN);
  for (int i = 62; i >= 0; i--) {
    f12_sh_cyclotomic_sqr_body(sh);
    if ((BLS_X_ABS >> i) & 1) fx_mul(sh, vp, lanes, t, 2, FX_PLAIN);
  }
}
// A <- conj(A)  (x < 0)
static __device__ __noinline__ void fx_conj(lds_u32* sh) {
#pragma unroll
  for (int j = 3; j < 6; j++) {
    fp x;
    sh_ld_fp(x, sh, 13 * j);
    fp_neg(x, x);
    fp_reduce(x, x);
    sh_st_fp(sh, 13 * j, x);
  }
}
// A <- A^x (x < 0), A in the cyclotomic subgroup
__device__ __forceinline__ void fx_pow_x(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t) {
  fx_store(sh, vp, lanes, t, 2);           // a: the fallback's operand
  fx_pow_run(sh, vp, lanes, t);
#if BLS_CYC_KARA
  if (fx_pow_finish_n<4, 0>(sh, vp, lanes, t)) {     // A = c = a^(2^61) conj(a^(2^57))
    fx_store(sh, vp, lanes, t, 6);
    for (int j = 0; j < 3; j++) fx_cyc_sqr(sh);
    fx_mul(sh, vp, lanes, t, 6, FX_CONJ);             // c^8 conj(c) = c^7
    fx_pow_finish_n<4, 1>(sh, vp, lanes, t);          // ... a^(2^48) a^(2^16)
  } else {
    fx_pow_plain(sh, vp, lanes, t);
  }
#else
  if (!fx_pow_finish(sh, vp, lanes, t)) fx_pow_plain(sh, vp, lanes, t);
#endif
  fx_conj(sh);
}
static __device__ __noinline__ int fx_is_one(lds_u32* sh) {
  fp12_t<hfp2> a;
  sh_ld_f12(a, sh);
  return fp12_is_one(a) ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}
// A == conj(V[slot]) ?
static __device__ __noinline__ int fx_eq_conj(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t, int slot) {
  const vs_ref v = {vp, lanes, t};
  bool eq = true;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    hfp2 a, b;
    sh_ld_fp(a.v, sh, 13 * j);
    vs_ld(b.v, v, slot, j);
    if (j >= 3) fp2_neg(b, b);
    eq = fp2_eq(a, b) && eq;
  }
  return eq ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}
// items [first, first + count) of the Fp12 workspace fws (stride n, as k_millerf2s / k_miller2s leave it)
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_finalexp2s(size_t n, size_t first, size_t count, const uint32_t* fws, uint32_t* vp, size_t lanes, int32_t* status) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t >> 1;
  if (j >= count) return;
  const size_t i = first + j;
  if (status[i] != BLS_OK) return;
  __shared__ uint32_t ash[F12_SH_WORDS * BLS_BLOCK];
  lds_u32* sh = lds_column(ash);
  {
    fp12_t<hfp2> f;
    ws_ld_hfp12(f, fws, n, i);
    fp12_reduce(f, f);       // the Miller kernel's conjugate carries negated limbs
    sh_st_f12(sh, f);
  }
  // easy part: f^((p^6 - 1)(p^2 + 1))
  fx_store(sh, vp, lanes, t, 1);                 // fin
  fx_inv(sh);
  fx_store(sh, vp, lanes, t, 0);                 // t = 1 / fin
  fx_load(sh, vp, lanes, t, 1, FX_CONJ);
  fx_mul(sh, vp, lanes, t, 0, FX_PLAIN);         // f = conj(fin) / fin
  fx_store(sh, vp, lanes, t, 1);
  fx_frob2(sh);
  fx_mul(sh, vp, lanes, t, 1, FX_PLAIN);         // f = f^(p^2) f
  fx_store(sh, vp, lanes, t, 1);                 // slot 1: f for the rest of the kernel
  // hard part: (x-1)^2 (x+p) (x^2+p^2-1) + 3
  fx_pow_x(sh, vp, lanes, t);
  fx_mul(sh, vp, lanes, t, 1, FX_CONJ);          // t = f^(x-1)
  fx_store(sh, vp, lanes, t, 0);
  fx_pow_x(sh, vp, lanes, t);
  fx_mul(sh, vp, lanes, t, 0, FX_CONJ);          // t = t^(x-1)
  fx_store(sh, vp, lanes, t, 0);
  fx_pow_x(sh, vp, lanes, t);
  fx_mul(sh, vp, lanes, t, 0, FX_FROB1);         // t = t^(x+p)
  fx_store(sh, vp, lanes, t, 0);
  fx_pow_x(sh, vp, lanes, t);
  fx_pow_x(sh, vp, lanes, t);                    // t^(x^2)
  fx_mul(sh, vp, lanes, t, 0, FX_FROB2);
  fx_mul(sh, vp, lanes, t, 0, FX_CONJ);          // t = t^(x^2+p^2-1)
  fx_store(sh, vp, lanes, t, 0);
  fx_load(sh, vp, lanes, t, 1, FX_PLAIN);
  fx_cyc_sqr(sh);
  fx_mul(sh, vp, lanes, t, 0, FX_PLAIN);         // t f^2
  const int st = fx_eq_conj(sh, vp, lanes, t, 1);   // == conj(f) = 1 / f  <=>  t f^3 == 1 (pairing.cuh final_exp_is_one)
  if (!lane_hi()) status[i] = st;
}
// The same final exponentiation in SEGMENTS between which k_cyc_run4 (four lanes per item, four waves per SIMD) runs the 63
// compressed squarings of each a^x: segment s ends by leaving a in slot 2, segment s + 1 starts by decompressing and multiplying
// the six saved powers (fx_pow_finish; the plain chain on the lanes of an item with a vanishing z2) and conjugating.
//   0: easy part, f -> slot 1, a = f          1: t = f^x conj(f) -> slot 0, a = t        2: t = t^x conj(t) -> slot 0, a = t
//   3: t = t^x t^p -> slot 0, a = t           4: a = t^x                                 5: t^(x^2) t^(p^2) conj(t) f^3 == 1 ?
__device__ __forceinline__ void fx_pow_end(lds_u32* sh, uint32_t* vp, size_t lanes, uint32_t t) {
  if (!fx_pow_finish(sh, vp, lanes, t)) fx_pow_plain(sh, vp, lanes, t);
  fx_conj(sh);
}
__global__ void __launch_bounds__(BLS_BLOCK, BLS_SPLIT_WAVES) __attribute__((disable_tail_calls))
k_finalexp_seg(int seg, size_t n, size_t first, size_t count, const uint32_t* fws, uint32_t* vp, size_t lanes, int32_t* status) {
  const uint32_t t = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t >> 1;
  if (j >= count) return;
  const size_t i = first + j;
  if (status[i] != BLS_OK) return;
  __shared__ uint32_t ash[F12_SH_WORDS * BLS_BLOCK];
  lds_u32* sh = lds_column(ash);
  if (seg == 0) {
    {
      fp12_t<hfp2> f;
      ws_ld_hfp12(f, fws, n, i);
      fp12_reduce(f, f);       // the Miller kernel's conjugate carries negated limbs
      sh_st_f12(sh, f);
    }
    fx_store(sh, vp, lanes, t, 1);                 // fin
    fx_inv(sh);
    fx_store(sh, vp, lanes, t, 0);                 // 1 / fin
    fx_load(sh, vp, lanes, t, 1, FX_CONJ);
    fx_mul(sh, vp, lanes, t, 0, FX_PLAIN);         // conj(fin) / fin
    fx_store(sh, vp, lanes, t, 1);
    fx_frob2(sh);
    fx_mul(sh, vp, lanes, t, 1, FX_PLAIN);         // f = f^(p^2) f
    fx_store(sh, vp, lanes, t, 1);                 // slot 1: f for the rest
    fx_store(sh, vp, lanes, t, 2);                 // a of the first a^x
    return;
  }
  fx_pow_end(sh, vp, lanes, t);
  if (seg == 1 || seg == 2) {
    fx_mul(sh, vp, lanes, t, seg == 1 ? 1 : 0, FX_CONJ);
  } else if (seg == 3) {
    fx_mul(sh, vp, lanes, t, 0, FX_FROB1);
  } else if (seg == 5) {
    fx_mul(sh, vp, lanes, t, 0, FX_FROB2);
    fx_mul(sh, vp, lanes, t, 0, FX_CONJ);          // t^(x^2+p^2-1)
    fx_store(sh, vp, lanes, t, 0);
    fx_load(sh, vp, lanes, t, 1, FX_PLAIN);
    fx_cyc_sqr(sh);
    fx_mul(sh, vp, lanes, t, 1, FX_PLAIN);         // f^3
    fx_mul(sh, vp, lanes, t, 0, FX_PLAIN);
    const int st = fx_is_one(sh);
    if (!lane_hi()) status[i] = st;
    return;
  }
  if (seg != 4) fx_store(sh, vp, lanes, t, 0);
  fx_store(sh, vp, lanes, t, 2);
}
#endif

#if defined(BLS_TU_CYCRUN4)
// =====================================================================================================
// The 63 compressed squarings of an a^x with FOUR lanes per item (round 3).  The running element is four Fp2 values
// (z2, z3, z4, z5) whose update is two independent Fp4 squarings (pairing.cuh cyc_c_sqr): lanes 0, 1 of a quad hold (z2, z3) --
// real parts on the even lane, imaginary parts on the odd lane, as everywhere in tower_split.cuh -- lanes 2, 3 hold (z4, z5);
// each pair squares its own Fp4 element and the results cross once per squaring by DPP quad_perm [2,3,0,1].  A lane's state is
// two Fp (26 packed LDS words) and three multiplier calls per squaring, so the kernel runs at FOUR waves per SIMD (128
// registers, 106 KB of LDS per CU): the multiply-add pipe is then saturated and the plain instructions between the multiplier
// calls issue at full rate, which two waves per SIMD cannot do (profiles/r03_ubench3_mad_rates.txt).  This loop is 60 % of the
// final exponentiation's instructions; nothing else on the path has a state this small.
// Store layout: as k_finalexp2s (value store, two lanes per item); a: slot 2, the six saved powers: cpow_slot().
#if !defined(BLS_TU_FINALEXP2)
#define VS_WORDS (6 * FP_NL)
__device__ __forceinline__ void cpow_slot(int k, int j, int& slot, int& idx) {   // Fp j (0..3: z2 z3 z4 z5) of saved power k (0..5)
  const int q = 4 * k + j;
  slot = 3 + q / 6;
  idx = q % 6;
}
#endif
__device__ __forceinline__ int32_t dpp_cross(int32_t x) { return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ void fp_cross(fp& r, const fp& a) {
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = dpp_cross(a.l[i]);
}
// 3 u + 2 z (plus) or 3 u - 2 z, reduced, into LDS word w0 -- tower_split.cuh cyc_store_plus / cyc_store_minus with the sign a lane value
__device__ __forceinline__ void cyc_store_signed(lds_u32* sh, int w0, const hfp2& u, const hfp2& z, bool plus) {
  hfp2 zz, nz, r;
  fp2_neg(nz, z);
  fp_sel(zz.v, plus, z.v, nz.v);
  fp2_add(r, u, zz);
  fp2_dbl(r, r);
  fp2_add(r, r, u);
  fp2_reduce(r, r);
  sh_st_fp(sh, w0, r.v);
}
// one compressed squaring on the two LDS values (words 0 and 13) of this lane; pair_b: this lane belongs to the (z4, z5) pair
static __device__ __noinline__ void cyc4_sqr_fn(lds_u32* sh, bool pair_b) {
  hfp2 a, b, t0, t1, r0, r1, x, u0, u1;
  sh_ld_fp(a.v, sh, 0);
  sh_ld_fp(b.v, sh, 13);
  fp4_sqr(t0, t1, a, b);
  fp_cross(r0.v, t0.v);                    // the other pair's (t_first, t_second)
  fp_cross(r1.v, t1.v);
  fp2_mul_xi(x, r1);                       // pair A needs xi t3 (pair B computes it too and drops it)
  fp2_norm(x, x);
  //   pair A (z2, z3):  z2' = 3 xi t3 + 2 z2,  z3' = 3 t2 - 2 z3         pair B (z4, z5):  z4' = 3 t0 - 2 z4,  z5' = 3 t1 + 2 z5
  fp_sel(u0.v, pair_b, r0.v, x.v);
  fp_sel(u1.v, pair_b, r1.v, r0.v);
  cyc_store_signed(sh, 0, u0, a, !pair_b);
  cyc_store_signed(sh, 13, u1, b, pair_b);
}
__global__ void __launch_bounds__(BLS_BLOCK, 4) __attribute__((disable_tail_calls))
k_cyc_run4(size_t count, uint32_t* vp, size_t lanes, const int32_t* status, size_t first) {
  const uint32_t t4 = blockIdx.x * BLS_BLOCK + threadIdx.x;
  const size_t j = t4 >> 2;
  if (j >= count) return;
  if (status[first + j] != BLS_OK) return;
  const bool pair_b = (t4 & 2u) != 0;
  const uint32_t t = (uint32_t)(2 * j) + (t4 & 1u);          // this lane's column of the (two lanes per item) value store
  __shared__ uint32_t csh[26 * BLS_BLOCK];
  lds_u32* sh = lds_column(csh);
  // tower order in a slot: c0.a0 c0.a1 c0.a2 c1.a0 c1.a1 c1.a2; z2 = c1.a0 (3), z3 = c0.a2 (2), z4 = c0.a1 (1), z5 = c1.a2 (5)
  const int i0 = pair_b ? 1 : 3, i1 = pair_b ? 5 : 2;
  {
    fp x;
    const uint32_t* row = vp + ((size_t)2 * VS_WORDS + (size_t)i0 * FP_NL) * lanes;
#pragma unroll
    for (int k = 0; k < FP_NL; k++) { x.l[k] = (int32_t)row[t]; row += lanes; }
    fp_reduce(x, x);
    sh_st_fp(sh, 0, x);
    row = vp + ((size_t)2 * VS_WORDS + (size_t)i1 * FP_NL) * lanes;
#pragma unroll
    for (int k = 0; k < FP_NL; k++) { x.l[k] = (int32_t)row[t]; row += lanes; }
    fp_reduce(x, x);
    sh_st_fp(sh, 13, x);
  }
  int saved = 0;
  for (int i = 1; i <= 63; i++) {
    cyc4_sqr_fn(sh, pair_b);
    if ((BLS_X_ABS >> i) & 1) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        fp x;
        int slot, idx;
        sh_ld_fp(x, sh, 13 * h);
        cpow_slot(saved, (pair_b ? 2 : 0) + h, slot, idx);
        uint32_t* row = vp + ((size_t)slot * VS_WORDS + (size_t)idx * FP_NL) * lanes;
#pragma unroll
        for (int k = 0; k < FP_NL; k++) { row[t] = (uint32_t)x.l[k]; row += lanes; }
      }
      saved++;
    }
  }
}
#endif

#if defined(BLS_TU_MSM1) || defined(BLS_TU_MSM2)
#include "msm2.cuh"
// =====================================================================================================
// Pippenger MSM: sum_i k_i P_i.  Scalars are < r < 2^255: W = floor(255 / c) windows, the first W - 1 are c bits wide
// (2^c buckets each) and the LAST takes the remaining clast = 255 - c (W - 1) bits (2^clast buckets), so no window is a
// narrow remainder whose few buckets would each receive n / 4 points.  Bucket b of window w sits at (w << c) + b.
//   k_msm_count / k_scan_* / k_msm_fill : counting sort of (window, digit) -> per-bucket index lists (multi-workgroup
//                    prefix sum of util_kernels.cuh)
//   k_msm_bucket   : one lane per bucket sums its points                     (n W / (1 - 2^-c) additions in total)
//   k_msm_chunk    : one lane per CH consecutive buckets of a window: running sums give sum_d (d - lo + 1) S_d, plus
//                    (lo - 1) * (chunk total) by a c-bit double-and-add, then c*w doublings weigh the window
//   k_point_fold   : tree sum of all chunk partials;  k_normalize: Z = 1 so the output bytes do not depend on the
//                    (atomic) fill order.
// Replaces the serial loop `aggregated_pk += pk.0 * *coeff` of reference src/secure_aggregation.rs:201-204.
__device__ __forceinline__ uint32_t msm_digit(const uint32_t* k, int bit, int c) {
  const int wi = bit >> 5, sh = bit & 31;
  uint64_t v = k[wi];
  if (wi + 1 < 8) v |= (uint64_t)k[wi + 1] << 32;
  return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}
#if defined(BLS_TU_MSM1)
__global__ void __launch_bounds__(BLS_BLOCK) k_msm_count(size_t n, const uint8_t* scalars, int c, int W, int clast, uint32_t* cnt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
  scalar_load_mod_r(k, scalars, i);
  for (int w = 0; w < W; w++) {
    uint32_t d = msm_digit(k, w * c, w == W - 1 ? clast : c);
    if (d) atomicAdd(&cnt[((size_t)w << c) + d], 1u);
  }
}
__global__ void __launch_bounds__(BLS_BLOCK) k_msm_fill(size_t n, const uint8_t* scalars, int c, int W, int clast, const uint32_t* off,
                                                      uint32_t* cursor, uint32_t* idx) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
  scalar_load_mod_r(k, scalars, i);
  for (int w = 0; w < W; w++) {
    uint32_t d = msm_digit(k, w * c, w == W - 1 ? clast : c);
    if (d) {
      size_t b = ((size_t)w << c) + d;
      uint32_t slot = atomicAdd(&cursor[b], 1u);
      idx[off[b] + slot] = (uint32_t)i;
    }
  }
}
#endif
template <int G>
struct msm_pt;
template <>
struct msm_pt<1> {
  typedef g1_jac jac_t;
  __device__ static void load(g1_jac& p, const uint8_t* b, size_t i, int fmt) { load_g1_pt(p, b, i, fmt); }
  __device__ static void store(uint8_t* b, size_t i, const g1_jac& p) { store_g1_pt(b, i, p); }
};
template <>
struct msm_pt<2> {
  typedef g2_jac jac_t;
  __device__ static void load(g2_jac& p, const uint8_t* b, size_t i, int fmt) { load_g2_pt(p, b, i, fmt); }
  __device__ static void store(uint8_t* b, size_t i, const g2_jac& p) { store_g2_pt(b, i, p); }
};
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_msm_bucket(size_t nb, const uint8_t* pts, int fmt, const uint32_t* perm, const uint32_t* cnt,
                                                        const uint32_t* off, const uint32_t* idx, uint8_t* sums) {
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  typename msm_pt<G>::jac_t acc, p;
  jac_set_inf(acc);
  const uint32_t n = cnt[b], o = off[b];
  for (uint32_t j = 0; j < n; j++) {
    uint32_t i = idx[o + j];
    msm_pt<G>::load(p, pts, perm ? perm[i] : i, fmt);
    jac_add(acc, acc, p);
  }
  msm_pt<G>::store(sums, b, acc);
}
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_msm_chunk(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials) {
  const size_t cpw = ((size_t)1 << c) / CH, cpl = ((size_t)1 << clast) / CH;   // chunks per regular / last window
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cpw * (W - 1) + cpl) return;
  const bool last = t >= cpw * (W - 1);
  const int w = last ? W - 1 : (int)(t / cpw);
  const size_t lo = (last ? t - cpw * (W - 1) : t % cpw) * CH;
  const int wbits = last ? clast : c;
  typename msm_pt<G>::jac_t run, acc, s;
  jac_set_inf(run);
  jac_set_inf(acc);
  for (int d = CH - 1; d >= 0; d--) {
    msm_pt<G>::load(s, sums, ((size_t)w << c) + lo + d, 0);
    jac_add(run, run, s);
    jac_add(acc, acc, run);   // acc = sum_d (d + 1) S_{lo + d}
  }
  // sum_d (lo + d) S = acc + (lo - 1) run
  if (lo == 0) {
    jac_neg(s, run);
  } else {
    jac_set_inf(s);
    const uint32_t mlt = (uint32_t)(lo - 1);
    for (int bit = wbits; bit >= 0; bit--) {
      jac_dbl(s, s);
      if ((mlt >> bit) & 1u) jac_add(s, s, run);
    }
  }
  jac_add(acc, acc, s);
  for (int k = 0; k < c * w; k++) jac_dbl(acc, acc);   // weight 2^(c w)
  msm_pt<G>::store(partials, t, acc);
}
// Z = 1 (or the canonical identity) in place
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_normalize(uint8_t* pt) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  typename msm_pt<G>::jac_t p;
  msm_pt<G>::load(p, pt, 0, 0);
  if (jac_is_inf(p)) {
    jac_set_inf(p);
  } else {
    decltype(p.x) zi, zi2;
    fe_inv(zi, p.z);
    fe_sqr(zi2, zi);
    fe_mul(p.x, p.x, zi2);
    fe_mul(zi2, zi2, zi);
    fe_mul(p.y, p.y, zi2);
    fe_one(p.z);
  }
  msm_pt<G>::store(pt, 0, p);
}
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_decompress(size_t n, const uint8_t* bytes, int legacy, uint8_t* out, int32_t* status, int keep) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep && status[i] != BLS_OK) return;
  typename msm_pt<G>::jac_t p;
  int rc;
  if (G == 1) rc = g1_decompress(*(g1_jac*)&p, bytes + 48 * i, legacy != 0);
  else rc = g2_decompress(*(g2_jac*)&p, bytes + 96 * i, legacy != 0);
  if (rc) jac_set_inf(p);
  msm_pt<G>::store(out, i, p);
  status[i] = rc;
}

// =====================================================================================================
// MSM, second generation (msm2.cuh).  Workspace of affine images: array of structs, entry e = i * E + j holds Q_j(P_i) in
// the internal limb form (G1: x, y = 28 words; G2: x.c0, x.c1, y.c0, y.c1 = 56 words): a bucket lane reads whole
// entries at random indices, so the struct is what it should find contiguous.
__device__ __forceinline__ void aff_st_fp(uint32_t* e, int w0, const fp& a) {
#pragma unroll
  for (int k = 0; k < FP_NL; k++) e[w0 + k] = (uint32_t)a.l[k];
}
__device__ __forceinline__ void aff_ld_fp(fp& r, const uint32_t* e, int w0) {
#pragma unroll
  for (int k = 0; k < FP_NL; k++) r.l[k] = (int32_t)e[w0 + k];
}
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_msm2_prep(size_t n, const uint8_t* pts, int fmt, const uint32_t* perm, uint32_t* affws, uint8_t* inf) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t src = perm ? perm[i] : i;
  typename msm_pt<G>::jac_t p;
  msm_pt<G>::load(p, pts, src, fmt);
  const bool isinf = jac_is_inf(p);
  inf[i] = isinf ? 1 : 0;
  if (isinf) return;
  uint32_t* e = affws + i * MSM2_E(G) * MSM2_AFF_WORDS(G);
  if (G == 1) {
    g1_aff a;
    jac_to_aff(a, *(g1_jac*)&p);
    fp qx[2], qy[2];
    msm2_images_g1(qx, qy, a);
    for (int j = 0; j < 2; j++) {
      aff_st_fp(e + j * 2 * FP_NL, 0, qx[j]);
      aff_st_fp(e + j * 2 * FP_NL, FP_NL, qy[j]);
    }
  } else {
    g2_aff a;
    jac_to_aff(a, *(g2_jac*)&p);
    fp2 qx[4], qy[4];
    msm2_images_g2(qx, qy, a);
    for (int j = 0; j < 4; j++) {
      uint32_t* q = e + j * 4 * FP_NL;
      aff_st_fp(q, 0, qx[j].c0);
      aff_st_fp(q, FP_NL, qx[j].c1);
      aff_st_fp(q, 2 * FP_NL, qy[j].c0);
      aff_st_fp(q, 3 * FP_NL, qy[j].c1);
    }
  }
}
// WEIGHTED TABLES (round 4; verify_secure only, where this runs while a host core still hashes the sorted key stream): for every key
// the multiples 2^start(w) P of its windows w = 1 .. W - 1, affine, each with its endomorphism images -- entry ((i E + j) W + w) of
// `tab` (entry w = 0: the images k_msm2_prep left).  A digit of window w then adds the ALREADY WEIGHTED point into its bucket, so the
// chunk lanes -- the latency chain of the whole sum -- lose their start(w) doublings (up to 52 on G2, 116 on G1) and the windows' sums
// are simply added.  Doubling commutes with the endomorphisms, so one chain of doublings per key serves all its images; the W - 1
// Jacobian multiples wait in `jt` (X, Y, Z and the running product of the Z's, 4 coordinates per window) for ONE shared inversion.
// Neither curve has a point of order two (both group orders are odd), so no multiple of a key is the identity and no Z vanishes.
__device__ __forceinline__ void co_st(uint32_t* e, const fp& a) { aff_st_fp(e, 0, a); }
__device__ __forceinline__ void co_st(uint32_t* e, const fp2& a) {
  aff_st_fp(e, 0, a.c0);
  aff_st_fp(e, FP_NL, a.c1);
}
__device__ __forceinline__ void co_ld(fp& r, const uint32_t* e) { aff_ld_fp(r, e, 0); }
__device__ __forceinline__ void co_ld(fp2& r, const uint32_t* e) {
  aff_ld_fp(r.c0, e, 0);
  aff_ld_fp(r.c1, e, FP_NL);
}
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_msm2_tables(size_t n, const uint32_t* affws, const uint8_t* inf, int W, uint32_t* tab, uint32_t* jt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || inf[i]) return;
  typedef typename msm_pt<G>::jac_t J;
  typedef decltype(J().x) F;
  constexpr int E = MSM2_E(G), AFFW = MSM2_AFF_WORDS(G), C = AFFW / 2;     // C: words per coordinate
  const msm2_layout L = msm2_make_layout(G == 1 ? 128 : 64, W);
  const uint32_t* e0 = affws + i * E * AFFW;
  uint32_t* ti = tab + i * E * (size_t)W * AFFW;
  for (int j = 0; j < E; j++)                                             // window 0: the images as they are
    for (int k = 0; k < AFFW; k++) ti[(size_t)j * W * AFFW + k] = e0[j * AFFW + k];
  if (W < 2) return;
  uint32_t* ji = jt + i * (size_t)(W - 1) * 4 * C;
  J T;
  co_ld(T.x, e0);
  co_ld(T.y, e0 + C);
  fe_one(T.z);
  F acc;
  for (int w = 1; w < W; w++) {
    const int k = msm2_start(L, w) - msm2_start(L, w - 1);
    for (int d = 0; d < k; d++) jac_dbl_body(T, T);     // inlined: through the non-inlined jac_dbl the point travelled by reference, through scratch
    uint32_t* r = ji + (size_t)(w - 1) * 4 * C;
    co_st(r, T.x);
    co_st(r + C, T.y);
    co_st(r + 2 * C, T.z);
    if (w == 1) acc = T.z;
    else fe_mul(acc, acc, T.z);
    co_st(r + 3 * C, acc);                                                // Z_1 ... Z_w
  }
  F inv;
  fe_inv(inv, acc);
  for (int w = W - 1; w >= 1; w--) {
    const uint32_t* r = ji + (size_t)(w - 1) * 4 * C;
    F zi, t, x, y;
    if (w > 1) {
      co_ld(t, r - 4 * C + 3 * C);                                        // Z_1 ... Z_(w-1)
      fe_mul(zi, inv, t);
      co_ld(t, r + 2 * C);
      fe_mul(inv, inv, t);
    } else {
      zi = inv;
    }
    fe_sqr(t, zi);
    co_ld(x, r);
    fe_mul(x, x, t);
    fe_mul(t, t, zi);
    co_ld(y, r + C);
    fe_mul(y, y, t);
    if (G == 1) {
      g1_aff a;
      a.x = *(fp*)&x;
      a.y = *(fp*)&y;
      a.inf = false;
      fp qx[2], qy[2];
      msm2_images_g1(qx, qy, a);
      for (int j = 0; j < 2; j++) {
        uint32_t* q = ti + ((size_t)j * W + w) * AFFW;
        aff_st_fp(q, 0, qx[j]);
        aff_st_fp(q, FP_NL, qy[j]);
      }
    } else {
      g2_aff a;
      a.x = *(fp2*)&x;
      a.y = *(fp2*)&y;
      a.inf = false;
      fp2 qx[4], qy[4];
      msm2_images_g2(qx, qy, a);
      for (int j = 0; j < 4; j++) {
        uint32_t* q = ti + ((size_t)j * W + w) * AFFW;
        aff_st_fp(q, 0, qx[j].c0);
        aff_st_fp(q, FP_NL, qx[j].c1);
        aff_st_fp(q, 2 * FP_NL, qy[j].c0);
        aff_st_fp(q, 3 * FP_NL, qy[j].c1);
      }
    }
  }
}
// scalar -> E sub-scalars (stored for k_msm2_fill) -> signed digits -> bucket sizes.  Bucket of digit magnitude m in
// window w: msm2_bucket_base(w) + m - 1 (window layout in msm2.cuh).
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_msm2_count(size_t n, const uint8_t* scalars, const uint8_t* inf, int W, uint32_t* cnt, uint64_t* subs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t a[4];
  if (G == 1) msm2_decompose_g1(a, (const uint32_t*)(scalars + 32 * i));
  else msm2_decompose_g2(a, (const uint32_t*)(scalars + 32 * i));
#pragma unroll
  for (int k = 0; k < 4; k++) subs[4 * i + k] = a[k];
  if (inf[i]) return;
  const int words = G == 1 ? 2 : 1;
  const msm2_layout L = msm2_make_layout(64 * words, W);
  for (int j = 0; j < MSM2_E(G); j++) {
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
      const int32_t d = msm2_digit(a + j * words, words, L, w, carry);
      if (d) atomicAdd(&cnt[msm2_bucket_base(L, w) + (size_t)((d < 0 ? -d : d) - 1)], 1u);
    }
  }
}
// bucket lists: entry code = (i * E + j) << 1 | (digit negative)
template <int G>
__global__ void __launch_bounds__(BLS_BLOCK) k_msm2_fill(size_t n, const uint64_t* subs, const uint8_t* inf, int W, const uint32_t* off,
                                                       uint32_t* cursor, uint32_t* idx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || inf[i]) return;
  uint64_t a[4];
#pragma unroll
  for (int k = 0; k < 4; k++) a[k] = subs[4 * i + k];
  const int words = G == 1 ? 2 : 1;
  const msm2_layout L = msm2_make_layout(64 * words, W);
  for (int j = 0; j < MSM2_E(G); j++) {
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
      const int32_t d = msm2_digit(a + j * words, words, L, w, carry);
      if (d) {
        const size_t b = msm2_bucket_base(L, w) + (size_t)((d < 0 ? -d : d) - 1);
        const uint32_t slot = atomicAdd(&cursor[b], 1u);
        idx[off[b] + slot] = ((uint32_t)(i * MSM2_E(G) + j) << 1) | (d < 0 ? 1u : 0u);
      }
    }
  }
}
// chunk t of the signed-digit layout -> (window, first bucket index inside the window)
__device__ __forceinline__ bool msm2_chunk_of(const msm2_layout& L, int CH, size_t t, int& w, size_t& lo) {
  const size_t cw_wide = ((size_t)1 << L.base) / CH, cw_narrow = ((size_t)1 << (L.base - 1)) / CH;
  const size_t nwide = cw_wide * L.rem;
  if (t < nwide) {
    w = (int)(t / cw_wide);
    lo = (t % cw_wide) * CH;
    return true;
  }
  t -= nwide;
  if (t >= cw_narrow * (L.W - L.rem)) return false;
  w = L.rem + (int)(t / cw_narrow);
  lo = (t % cw_narrow) * CH;
  return true;
}
#if defined(BLS_TU_MSM2)
// k_msm_chunk for G2 on TWO lanes per chunk (jac<hfp2>, tower_split.cuh): the 2^(c w) doublings are the critical path of
// the chunk lanes, and a lane-split Fp2 doubling costs half the instructions per lane.  Each lane converts and keeps its
// own component (real / imaginary) of every coordinate.
// k_msm_bucket for G2 on two lanes per bucket
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_msm_bucket_g2s(size_t nb, const uint8_t* pts, int fmt, const uint32_t* perm, const uint32_t* cnt,
                                                            const uint32_t* off, const uint32_t* idx, uint8_t* sums) {
  const size_t b = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (b >= nb) return;
  jac<hfp2> acc, p;
  jac_set_inf(acc);
  const uint32_t n = cnt[b], o = off[b];
  for (uint32_t j = 0; j < n; j++) {
    const uint32_t i = idx[o + j];
    ld_g2s_fmt(p, pts, perm ? perm[i] : i, fmt);
    jac_add(acc, acc, p);
  }
  st_g2s(sums, b, acc);
}
// One G2 doubling by the FOUR lanes of a chunk -- two lane pairs that hold the same lane-split point: the schedule of
// jac_dbl_pair (h2c.cuh) one level up, lane PAIR 0 taking A = X^2, F = (3A)^2, YZ, E (D - X3) and lane pair 1 B = Y^2,
// C = B^2, (X + B)^2; the pairs exchange by DPP quad_perm [2,3,0,1].  Same operation order and reductions as jac_dbl_body.
__device__ __forceinline__ void hfp2_swap2(hfp2& r, const hfp2& a) {
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.v.l[i] = __builtin_amdgcn_mov_dpp(a.v.l[i], 0x4E, 0xF, 0xF, true);
}
__device__ __forceinline__ void hfp2_sel(hfp2& r, bool c, const hfp2& a, const hfp2& b) { fp_sel(r.v, c, a.v, b.v); }
__device__ __forceinline__ void jac_dbl_quad(jac<hfp2>& p, bool hi2) {
  hfp2 s1, s2, s3, in, a, b, o2, o3, D, E, t, x3, y3, z3, c8;
  hfp2_sel(in, hi2, p.y, p.x);
  fp2_sqr(s1, in);                // pair 0: A = X^2          pair 1: B = Y^2
  fp2_dbl(E, s1);
  fp2_add(E, E, s1);
  fp2_reduce(E, E);               // pair 0: E = 3A
  hfp2_sel(in, hi2, s1, E);
  fp2_sqr(s2, in);                // pair 0: F = E^2          pair 1: C = B^2
  fp2_add(t, p.x, s1);
  fp2_norm(t, t);                 //                          pair 1: X + B
  hfp2_sel(a, hi2, t, p.y);
  hfp2_sel(b, hi2, t, p.z);
  fp2_mul(s3, a, b);              // pair 0: Y Z              pair 1: (X + B)^2
  hfp2_swap2(o2, s2);             // pair 0: C
  hfp2_swap2(o3, s3);             // pair 0: (X + B)^2
  fp2_sub(t, o3, s1);
  fp2_sub(t, t, o2);
  fp2_dbl(D, t);
  fp2_reduce(D, D);               // D = 2((X + B)^2 - A - C)
  fp2_dbl(t, D);
  fp2_sub(t, s2, t);
  fp2_reduce(x3, t);              // X3 = F - 2D
  fp2_sub(t, D, x3);
  fp2_norm(t, t);
  fp2_mul(t, E, t);               // E (D - X3)
  fp2_dbl(c8, o2);
  fp2_dbl(c8, c8);
  fp2_reduce(c8, c8);
  fp2_dbl(c8, c8);                // 8C
  fp2_sub(t, t, c8);
  fp2_reduce(y3, t);
  fp2_dbl(z3, s3);
  fp2_reduce(z3, z3);             // Z3 = 2YZ
  hfp2_swap2(a, x3);
  hfp2_swap2(b, y3);
  hfp2_swap2(t, z3);
  hfp2_sel(p.x, hi2, a, x3);
  hfp2_sel(p.y, hi2, b, y3);
  hfp2_sel(p.z, hi2, t, z3);
}
// k_msm_chunk for G2 on FOUR lanes per chunk: as k_msm_chunk_g2s, the doubling chains (up to 240 doublings for the window
// weight: the critical path of the MSM) shared by the two lane pairs
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_msm_chunk_g2q(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials) {
  const size_t cpw = ((size_t)1 << c) / CH, cpl = ((size_t)1 << clast) / CH;   // chunks per regular / last window
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = gid >> 2;
  const bool hi2 = ((gid >> 1) & 1) != 0;
  if (t >= cpw * (W - 1) + cpl) return;
  const bool last = t >= cpw * (W - 1);
  const int w = last ? W - 1 : (int)(t / cpw);
  const size_t lo = (last ? t - cpw * (W - 1) : t % cpw) * CH;
  const int wbits = last ? clast : c;
  jac<hfp2> run, acc, s;
  jac_set_inf(run);
  jac_set_inf(acc);
  for (int d = CH - 1; d >= 0; d--) {
    ld_g2s(s, sums, ((size_t)w << c) + lo + d);
    jac_add(run, run, s);
    jac_add(acc, acc, run);   // acc = sum_d (d + 1) S_{lo + d}
  }
  if (lo == 0) {              // sum_d (lo + d) S = acc + (lo - 1) run
    jac_neg(s, run);
  } else {
    jac_set_inf(s);
    const uint32_t mlt = (uint32_t)(lo - 1);
    for (int bit = wbits; bit >= 0; bit--) {
      jac_dbl_quad(s, hi2);
      if ((mlt >> bit) & 1u) jac_add(s, s, run);
    }
  }
  jac_add(acc, acc, s);
  for (int k = 0; k < c * w; k++) jac_dbl_quad(acc, hi2);   // weight 2^(c w)
  if (!hi2) st_g2s(partials, t, acc);
}
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_msm_chunk_g2s(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials) {
  const size_t cpw = ((size_t)1 << c) / CH, cpl = ((size_t)1 << clast) / CH;   // chunks per regular / last window
  const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (t >= cpw * (W - 1) + cpl) return;
  const bool last = t >= cpw * (W - 1);
  const int w = last ? W - 1 : (int)(t / cpw);
  const size_t lo = (last ? t - cpw * (W - 1) : t % cpw) * CH;
  const int wbits = last ? clast : c;
  jac<hfp2> run, acc, s;
  jac_set_inf(run);
  jac_set_inf(acc);
  for (int d = CH - 1; d >= 0; d--) {
    ld_g2s(s, sums, ((size_t)w << c) + lo + d);
    jac_add(run, run, s);
    jac_add(acc, acc, run);   // acc = sum_d (d + 1) S_{lo + d}
  }
  if (lo == 0) {              // sum_d (lo + d) S = acc + (lo - 1) run
    jac_neg(s, run);
  } else {
    jac_set_inf(s);
    const uint32_t mlt = (uint32_t)(lo - 1);
    for (int bit = wbits; bit >= 0; bit--) {
      jac_dbl(s, s);
      if ((mlt >> bit) & 1u) jac_add(s, s, run);
    }
  }
  jac_add(acc, acc, s);
  for (int k = 0; k < c * w; k++) jac_dbl(acc, acc);   // weight 2^(c w)
  st_g2s(partials, t, acc);
}
#endif
#if defined(BLS_TU_MSM1)
// k_msm_chunk for G1 on TWO lanes per chunk: everything runs on both lanes with identical operands except the doubling
// chains -- the c-bit multiplier and, above all, the c * w doublings of the window weight, up to 240 in a row and the
// critical path of the whole MSM (the kernel is latency-bound: a few hundred waves) -- which use the shared two-lane
// doubling of h2c.cuh (four steps instead of seven products).
__global__ void __launch_bounds__(BLS_BLOCK) k_msm_chunk_g1p(int c, int W, int clast, int CH, const uint8_t* sums, uint8_t* partials) {
  const size_t cpw = ((size_t)1 << c) / CH, cpl = ((size_t)1 << clast) / CH;   // chunks per regular / last window
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = gid >> 1;
  const bool hi = (gid & 1) != 0;
  if (t >= cpw * (W - 1) + cpl) return;
  const bool last = t >= cpw * (W - 1);
  const int w = last ? W - 1 : (int)(t / cpw);
  const size_t lo = (last ? t - cpw * (W - 1) : t % cpw) * CH;
  const int wbits = last ? clast : c;
  g1_jac run, acc, s;
  jac_set_inf(run);
  jac_set_inf(acc);
  for (int d = CH - 1; d >= 0; d--) {
    load_g1_pt(s, sums, ((size_t)w << c) + lo + d, 0);
    jac_add(run, run, s);
    jac_add(acc, acc, run);   // acc = sum_d (d + 1) S_{lo + d}
  }
  if (lo == 0) {              // sum_d (lo + d) S = acc + (lo - 1) run
    jac_neg(s, run);
  } else {
    jac_set_inf(s);
    const uint32_t mlt = (uint32_t)(lo - 1);
    for (int bit = wbits; bit >= 0; bit--) {
      jac_dbl_pair(s, hi);
      if ((mlt >> bit) & 1u) jac_add(s, s, run);
    }
  }
  jac_add(acc, acc, s);
  for (int k = 0; k < c * w; k++) jac_dbl_pair(acc, hi);   // weight 2^(c w)
  if (!hi) store_g1_pt(partials, t, acc);
}

// one lane per (bucket, part): part q of Q takes entries q, q + Q, ... of the bucket's list (Q > 1 fills the machine when
// there are fewer buckets than lanes); mixed additions of affine images
// tabW != 0: affws is the weighted table of k_msm2_tables (tabW windows per image): the entry of this bucket's window
__global__ void __launch_bounds__(BLS_BLOCK) k_msm2_bucket_g1(size_t nb, int Q, const uint32_t* affws, const uint32_t* cnt, const uint32_t* off,
                                                           const uint32_t* idx, uint8_t* sums, int tabW) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * (size_t)Q) return;
  const size_t b = t / Q;
  const uint32_t q = (uint32_t)(t % Q);
  g1_jac acc;
  jac_set_inf(acc);
  const uint32_t m = cnt[b], o = off[b];
  const size_t per = tabW ? (size_t)tabW : 1, wsel = tabW ? (size_t)msm2_window_of(msm2_make_layout(128, tabW), b) : 0;
  for (uint32_t j = q; j < m; j += Q) {
    const uint32_t code = idx[o + j];
    const uint32_t* e = affws + ((size_t)(code >> 1) * per + wsel) * 2 * FP_NL;
    fp x, y;
    aff_ld_fp(x, e, 0);
    aff_ld_fp(y, e, FP_NL);
    if (code & 1u) fp_neg(y, y);
    jac_madd_body(acc, acc, x, y);
  }
  store_g1_pt(sums, t, acc);
}
// the Q parts of every bucket added up in place (part 0 keeps the bucket's sum): one lane per bucket, fully parallel,
// so that the latency-bound chunk lanes below read one sum per bucket
__global__ void __launch_bounds__(BLS_BLOCK) k_msm2_merge_g1(size_t nb, int Q, uint8_t* sums) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  g1_jac acc, s;
  load_g1_pt(acc, sums, b * Q, 0);
  for (int q = 1; q < Q; q++) {
    load_g1_pt(s, sums, b * Q + q, 0);
    jac_add_body(acc, acc, s);
  }
  store_g1_pt(sums, b * Q, acc);
}
// chunk lanes as k_msm_chunk_g1p (two lanes per chunk sharing the doubling chains), for the signed-digit layout: bucket index
// t of a window stands for digit t + 1 and window w weighs 2^start(w); bucket sums sit `stride` records apart
// weighted != 0: the bucket sums already carry their window's weight (k_msm2_tables): no doublings at the end
__global__ void __launch_bounds__(BLS_BLOCK) k_msm2_chunk_g1p(int W, int CH, int stride, const uint8_t* sums, uint8_t* partials, int weighted) {
  const msm2_layout L = msm2_make_layout(128, W);
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = gid >> 1;
  const bool hi = (gid & 1) != 0;
  int w;
  size_t lo;
  if (!msm2_chunk_of(L, CH, t, w, lo)) return;
  g1_jac run, acc, s;
  jac_set_inf(run);
  jac_set_inf(acc);
  for (int d = CH - 1; d >= 0; d--) {
    load_g1_pt(s, sums, (msm2_bucket_base(L, w) + lo + d) * stride, 0);
    jac_add_body(run, run, s);
    jac_add_body(acc, acc, run);   // acc = sum_d (d + 1) S_{lo + d}
  }
  if (lo != 0) {              // sum_d (lo + d + 1) S = acc + lo * run
    jac_set_inf(s);
    const uint32_t mlt = (uint32_t)lo;
    for (int bit = msm2_width(L, w) - 2; bit >= 0; bit--) {
      jac_dbl_pair(s, hi);
      if ((mlt >> bit) & 1u) jac_add(s, s, run);
    }
    jac_add_body(acc, acc, s);
  }
  const int shift = weighted ? 0 : msm2_start(L, w);
  for (int k = 0; k < shift; k++) jac_dbl_pair(acc, hi);   // the window's weight
  if (!hi) store_g1_pt(partials, t, acc);
}
template __global__ void k_msm2_prep<1>(size_t, const uint8_t*, int, const uint32_t*, uint32_t*, uint8_t*);
template __global__ void k_msm2_tables<1>(size_t, const uint32_t*, const uint8_t*, int, uint32_t*, uint32_t*);
template __global__ void k_msm2_count<1>(size_t, const uint8_t*, const uint8_t*, int, uint32_t*, uint64_t*);
template __global__ void k_msm2_fill<1>(size_t, const uint64_t*, const uint8_t*, int, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_decompress<1>(size_t, const uint8_t*, int, uint8_t*, int32_t*, int);
template __global__ void k_msm_bucket<1>(size_t, const uint8_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint8_t*);
template __global__ void k_msm_chunk<1>(int, int, int, int, const uint8_t*, uint8_t*);
template __global__ void k_normalize<1>(uint8_t*);
#else

// G2 buckets on two lanes per (bucket, part) (jac<hfp2>): each lane loads its own component of the affine entry
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_msm2_bucket_g2s(size_t nb, int Q, const uint32_t* affws, const uint32_t* cnt, const uint32_t* off,
                                                               const uint32_t* idx, uint8_t* sums, int tabW) {
  const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (t >= nb * (size_t)Q) return;
  const size_t b = t / Q;
  const uint32_t q = (uint32_t)(t % Q);
  jac<hfp2> acc;
  jac_set_inf(acc);
  const uint32_t m = cnt[b], o = off[b];
  const int half = lane_hi() ? FP_NL : 0;
  const size_t per = tabW ? (size_t)tabW : 1, wsel = tabW ? (size_t)msm2_window_of(msm2_make_layout(64, tabW), b) : 0;
  for (uint32_t j = q; j < m; j += Q) {
    const uint32_t code = idx[o + j];
    const uint32_t* e = affws + ((size_t)(code >> 1) * per + wsel) * 4 * FP_NL;
    hfp2 x, y;
    aff_ld_fp(x.v, e, half);
    aff_ld_fp(y.v, e, 2 * FP_NL + half);
    if (code & 1u) fp_neg(y.v, y.v);
    jac_madd_body(acc, acc, x, y);
  }
  st_g2s(sums, t, acc);
}
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_msm2_merge_g2s(size_t nb, int Q, uint8_t* sums) {
  const size_t b = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (b >= nb) return;
  jac<hfp2> acc, s;
  ld_g2s(acc, sums, b * Q);
  for (int q = 1; q < Q; q++) {
    ld_g2s(s, sums, b * Q + q);
    jac_add_body(acc, acc, s);
  }
  st_g2s(sums, b * Q, acc);
}
// chunk lanes as k_msm_chunk_g2q (four lanes per chunk: two lane pairs sharing the doubling chains), signed-digit layout
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_msm2_chunk_g2q(int W, int CH, int stride, const uint8_t* sums, uint8_t* partials, int weighted) {
  const msm2_layout L = msm2_make_layout(64, W);
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = gid >> 2;
  const bool hi2 = ((gid >> 1) & 1) != 0;
  int w;
  size_t lo;
  if (!msm2_chunk_of(L, CH, t, w, lo)) return;
  jac<hfp2> run, acc, s;
  jac_set_inf(run);
  jac_set_inf(acc);
  for (int d = CH - 1; d >= 0; d--) {
    ld_g2s(s, sums, (msm2_bucket_base(L, w) + lo + d) * stride);
    jac_add_body(run, run, s);
    jac_add_body(acc, acc, run);
  }
  if (lo != 0) {
    jac_set_inf(s);
    const uint32_t mlt = (uint32_t)lo;
    for (int bit = msm2_width(L, w) - 2; bit >= 0; bit--) {
      jac_dbl_quad(s, hi2);
      if ((mlt >> bit) & 1u) jac_add(s, s, run);
    }
    jac_add_body(acc, acc, s);
  }
  const int shift = weighted ? 0 : msm2_start(L, w);
  for (int k = 0; k < shift; k++) jac_dbl_quad(acc, hi2);
  if (!hi2) st_g2s(partials, t, acc);
}
template __global__ void k_msm2_prep<2>(size_t, const uint8_t*, int, const uint32_t*, uint32_t*, uint8_t*);
template __global__ void k_msm2_tables<2>(size_t, const uint32_t*, const uint8_t*, int, uint32_t*, uint32_t*);
template __global__ void k_msm2_count<2>(size_t, const uint8_t*, const uint8_t*, int, uint32_t*, uint64_t*);
template __global__ void k_msm2_fill<2>(size_t, const uint64_t*, const uint8_t*, int, const uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_decompress<2>(size_t, const uint8_t*, int, uint8_t*, int32_t*, int);
template __global__ void k_msm_bucket<2>(size_t, const uint8_t*, int, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint8_t*);
template __global__ void k_msm_chunk<2>(int, int, int, int, const uint8_t*, uint8_t*);
template __global__ void k_normalize<2>(uint8_t*);
#endif
#endif  // BLS_TU_MSM*

#if defined(BLS_TU_COOP)
// =====================================================================================================
#include "coop.cuh"
// Miller loop of the item's two pairs + final exponentiation + verdict, one workgroup (= one wave) per item
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_pairing_coop(size_t n, const uint32_t* pairs, int32_t* status, int fixed_g2) {
  __shared__ coop_shared S;
  const size_t i = blockIdx.x;
  if (i >= n) return;
  if (status[i] != BLS_OK) return;
  g1_aff P[2];
  aff<hfp2> Q[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int w0 = k * 3 * W2;
    ws_ld_fp(P[k].x, pairs, n, i, w0);
    ws_ld_fp(P[k].y, pairs, n, i, w0 + W1);
    ws_ld_fp(Q[k].x.v, pairs, n, i, w0 + W2 + (lane_hi() ? W1 : 0));
    ws_ld_fp(Q[k].y.v, pairs, n, i, w0 + 2 * W2 + (lane_hi() ? W1 : 0));
    P[k].inf = false;
    Q[k].inf = false;
  }
  coop_miller2(S, P, Q, fixed_g2);
  int st = coop_final_verdict(S);
  if (threadIdx.x == 0) status[i] = st;
}
// final exponentiation + verdict of workspace item 0
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_finalexp_coop(const uint32_t* fws, size_t stride, int32_t* verdict) {
  __shared__ coop_shared S;
  if (blockIdx.x != 0) return;
  // workspace order is the tower order c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2 -> powers 0, 2, 4, 1, 3, 5
  const int k = coop_pair();
  if (k < 6) {
    const int pw = (k < 3) ? 2 * k : 2 * (k - 3) + 1;
    hfp2 x;
    ws_ld_fp(x.v, fws, stride, 0, W2 * k + (lane_hi() ? W1 : 0));
    coop_st(S.f.c[pw], x);
  }
  __syncthreads();
  int st = coop_final_verdict(S);
  if (threadIdx.x == 0) *verdict = st;
}
// Miller loop of the item's two pairs and the EASY part of the final exponentiation, f^((p^6 - 1)(p^2 + 1)), exported in the
// value layout of the row-wide engine (coefficient k of the basis w^0..w^5, real then imaginary part, sixteen words each):
// the hard part follows in k_finalexp_wide
__global__ void __launch_bounds__(BLS_BLOCK, 2) k_pairing_coop_easy(size_t n, const uint32_t* pairs, const int32_t* status, int fixed_g2, uint32_t* easy) {
  __shared__ coop_shared S;
  const size_t i = blockIdx.x;
  if (i >= n) return;
  if (status[i] != BLS_OK) return;
  g1_aff P[2];
  aff<hfp2> Q[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int w0 = k * 3 * W2;
    ws_ld_fp(P[k].x, pairs, n, i, w0);
    ws_ld_fp(P[k].y, pairs, n, i, w0 + W1);
    ws_ld_fp(Q[k].x.v, pairs, n, i, w0 + W2 + (lane_hi() ? W1 : 0));
    ws_ld_fp(Q[k].y.v, pairs, n, i, w0 + 2 * W2 + (lane_hi() ? W1 : 0));
    P[k].inf = false;
    Q[k].inf = false;
  }
  coop_miller2(S, P, Q, fixed_g2);
  coop_final_easy(S);
  uint32_t* e = easy + i * WIDE_EASY_WORDS;
  for (int t = threadIdx.x; t < WIDE_EASY_WORDS; t += BLS_BLOCK) {
    const int v = t >> 4, l = t & 15;                  // value v = 2 k + component
    e[t] = l < FP_NL ? S.f.c[v >> 1][(v & 1) * FP_NL + l] : 0u;
  }
}
#endif

#if defined(BLS_TU_WIDE)
// =====================================================================================================
#include "wide.cuh"
#include "wide_fp2.cuh"
// self-test of the row-wide multiplier: item i = a_i * b_i (caller format: 48-byte Montgomery words), `reps` chained
// multiplications by b (reps = 1: the plain product), four items per wave
__global__ void __launch_bounds__(WIDE_BLOCK) k_wide_mul_test(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int reps) {
  __shared__ uint32_t sh[WIDE_BLOCK / 16][3][16];
  const int row = threadIdx.x >> 4, l = threadIdx.x & 15;
  const size_t i = (size_t)blockIdx.x * (blockDim.x >> 4) + row;
  wide_consts K;
  wide_init(K);
  if (l == 0 && i < n) {
    fp x, y;
    fp_from_raw(x, (const uint32_t*)(a + 48 * i));
    fp_from_raw(y, (const uint32_t*)(b + 48 * i));
    w_store_local(sh[row][0], x);
    w_store_local(sh[row][1], y);
  }
  __syncthreads();
  wfp wa = i < n ? (wfp)sh[row][0][l] : 0, wb = i < n ? (wfp)sh[row][1][l] : 0;
  for (int r = 0; r < reps; r++) wa = w_mul(wa, wb, K);
  sh[row][2][l] = (uint32_t)wa;
  __syncthreads();
  if (l == 0 && i < n) {
    fp z;
    w_load_local(z, sh[row][2]);
    fp_to_raw((uint32_t*)(out + 48 * i), z);
  }
}
#include "wide_engine.cuh"
bool wide_prog_is_fp12(const uint32_t* prog, size_t len) {
  if (len > WIDE_PROG_MAX) return false;
  for (size_t k = 0; k < len; k++) {
    const uint32_t w0 = prog[2 * k], w1 = prog[2 * k + 1], op = w0 & 0xffffu;
    const uint32_t refs[3] = {w0 >> 16, w1 & 0xffffu, w1 >> 16};
    if (op == WOP_FPINV) {                      // single values anywhere in the five Fp12 arrays
      for (uint32_t r : refs)
        if (r < WV_F || r >= WV_ACC + 12) return false;
      continue;
    }
    if (op >= WOP_PDBL1 && op != WOP_MUL_LINE && op != WOP_CYC_SQRC) return false;
    for (uint32_t r : refs)
      if (r != WV_F && r != WV_T && r != WV_U && r != WV_W && r != WV_ACC) return false;
  }
  return true;
}
// self-test / measurement: a caller-supplied program on one workgroup.  F, U, W, ACC <- the Fp12 at fin (twelve 48-byte
// Montgomery elements), T <- 0; the program runs `reps` times; tout <- T.  (The host entry validates the program words.)
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_wide_prog_test(const uint32_t* prog, int len, int reps, const uint8_t* fin, uint8_t* tout) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  wide_consts K;
  wide_init(K);
  wide_stage(S, prog, len);
  if (threadIdx.x < 12) {
    fp x;
    fp_from_raw(x, (const uint32_t*)(fin + 48 * threadIdx.x));
    w_store_local(S.V[WV_F + threadIdx.x], x);
    w_store_local(S.V[WV_U + threadIdx.x], x);
    w_store_local(S.V[WV_W + threadIdx.x], x);
    w_store_local(S.V[WV_ACC + threadIdx.x], x);
    fp_zero(x);
    w_store_local(S.V[WV_T + threadIdx.x], x);
  }
  __syncthreads();
  for (int r = 0; r < reps; r++) wide_exec(S, len, K);
  if (threadIdx.x < 12) {
    fp x;
    w_load_local(x, S.V[WV_T + threadIdx.x]);
    fp_to_raw((uint32_t*)(tout + 48 * threadIdx.x), x);
  }
}
// the hard part of the final exponentiation and the comparison with one, one 256-thread workgroup per item
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_finalexp_wide(size_t n, const uint32_t* easy, int32_t* status) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t item = blockIdx.x;
  if (item >= n) return;
  if (status[item] != BLS_OK) return;                   // uniform over the workgroup
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_FINAL_HARD, WIDE_PROG_FINAL_HARD_LEN);
  for (int t = threadIdx.x; t < WIDE_EASY_WORDS; t += WIDE_ENGINE_BLOCK) S.V[WV_F + (t >> 4)][t & 15] = easy[item * WIDE_EASY_WORDS + t];
  if (threadIdx.x == 0) S.flag = 1;
  __syncthreads();
  wide_exec(S, WIDE_PROG_FINAL_HARD_LEN, K);
  // == 1 ?  (lane-local canonical comparison of the twelve components)
  if (threadIdx.x < 12) {
    fp x, one;
    w_load_local(x, S.V[WV_T + threadIdx.x]);
    bool ok;
    if (threadIdx.x == 0) {
      fp_one(one);
      ok = fp_eq(x, one);
    } else {
      ok = fp_is_zero(x);
    }
    if (!ok) S.flag = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) status[item] = S.flag ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}

// The whole two-pair pairing check of one item on the row-wide engine: the line coefficients of the variable G2 arguments,
// the Miller loop, the final exponentiation and the comparison with one (program PAIR_FIXED when pair 1's G2 argument is
// the constant -g2, whose lines come from the table G2NEG_LINES; PAIR_GENERAL otherwise).  pairs: the affine workspace that
// the prepare stage writes (word-major: word k of item i at pairs[k n + i]).
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_pairing_wide(size_t n, const uint32_t* pairs, int32_t* status, int fixed_g2) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t item = blockIdx.x;
  if (item >= n) return;
  if (status[item] != BLS_OK) return;                   // uniform over the workgroup
  wide_consts K;
  wide_init(K);
  if (fixed_g2) wide_stage(S, WIDE_PROG_PAIR_FIXED, WIDE_PROG_PAIR_FIXED_LEN);
  else wide_stage(S, WIDE_PROG_PAIR_GENERAL, WIDE_PROG_PAIR_GENERAL_LEN);
  const int row = (int)(threadIdx.x >> 4), l = (int)(threadIdx.x & 15u);
  auto ws_word = [&](int w) -> uint32_t { return l < FP_NL ? pairs[(size_t)(w + l) * n + item] : 0u; };
  // per pair k: P = (x, y) at words w0, w0 + W1; Q = (x.c0, x.c1, y.c0, y.c1) at w0 + W2 ...
  if (row < 2) {
    const int w0 = row * 3 * W2;
    S.V[WV_P + 4 * row][l] = ws_word(w0);             // the G1 point as the Jacobian triple (x, y, 1): the programs take any Z
    S.V[WV_P + 4 * row + 1][l] = ws_word(w0 + W1);
    S.V[WV_P + 4 * row + 2][l] = l < FP_NL ? FP_ONE[l] : 0u;
    S.V[WV_P + 4 * row + 3][l] = 0u;
    const uint32_t pt = row == 0 ? WV_PT0 : WV_PT1;
    for (int c4 = 0; c4 < 4; c4++) S.V[pt + 6 + c4][l] = ws_word(w0 + W2 + c4 * W1);   // Q = (x, y, 1), Jacobian as the programs expect
    S.V[pt + 10][l] = l < FP_NL ? FP_ONE[l] : 0u;
    S.V[pt + 11][l] = 0u;
  }
  if (row == 2) {
    S.V[WV_F][l] = l < FP_NL ? FP_ONE[l] : 0u;     // f = 1
    for (int v = 1; v < 12; v++) S.V[WV_F + v][l] = 0u;
  }
  if (fixed_g2) {                                    // pair 1's unscaled lines: the precomputed table of -g2 (2: of -[c] g2)
    const uint32_t (*lines)[6 * FP_NL] = fixed_g2 == 2 ? G2NEGC_LINES : G2NEG_LINES;
    for (int t = threadIdx.x; t < WIDE_STEPS * 6 * 16; t += WIDE_ENGINE_BLOCK) {
      const int st = t / 96, v = (t % 96) >> 4, ll = t & 15;
      S.V[WV_L + 12 * st + 6 + v][ll] = ll < FP_NL ? lines[st][v * FP_NL + ll] : 0u;
    }
  }
  if (threadIdx.x == 0) S.flag = 1;
  __syncthreads();
  wide_exec(S, fixed_g2 ? WIDE_PROG_PAIR_FIXED_LEN : WIDE_PROG_PAIR_GENERAL_LEN, K);
  if (threadIdx.x < 12) {
    fp x, one;
    w_load_local(x, S.V[WV_T + threadIdx.x]);
    bool ok;
    if (threadIdx.x == 0) {
      fp_one(one);
      ok = fp_eq(x, one);
    } else {
      ok = fp_is_zero(x);
    }
    if (!ok) S.flag = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) status[item] = S.flag ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}

// the whole final exponentiation of item 0 of an Fp12 workspace (the Miller product of an aggregate verification) and the
// comparison with one: program FINAL on one workgroup (0.65 ms against 1.3 ms for the wave-cooperative form)
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_finalexp_wide_ws(const uint32_t* fws, size_t stride, int32_t* verdict) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  if (blockIdx.x != 0) return;
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_FINAL, WIDE_PROG_FINAL_LEN);
  const int v = (int)(threadIdx.x >> 4), l = (int)(threadIdx.x & 15u);
  if (v < 12) {   // workspace order is the tower order c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2 -> powers 0, 2, 4, 1, 3, 5 of w
    const int k = v >> 1, pw = k < 3 ? 2 * k : 2 * (k - 3) + 1;
    S.V[WV_F + 2 * pw + (v & 1)][l] = l < FP_NL ? fws[(size_t)(W1 * v + l) * stride] : 0u;
  }
  if (threadIdx.x == 0) S.flag = 1;
  __syncthreads();
  wide_exec(S, WIDE_PROG_FINAL_LEN, K);
  if (threadIdx.x < 12) {
    fp x, one;
    w_load_local(x, S.V[WV_T + threadIdx.x]);
    bool ok;
    if (threadIdx.x == 0) {
      fp_one(one);
      ok = fp_eq(x, one);
    } else {
      ok = fp_is_zero(x);
    }
    if (!ok) S.flag = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) *verdict = S.flag ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}
// The last levels of a pairing product's fold tree on the engine (round 3): workgroup b <- the product of the Fp12 items
// [16 b, 16 b + 16) of a workspace (absent ones count as 1), written as item b of another workspace.  Fifteen general products of
// ~2.8 us against one 0.15 ms launch per halving with the lane-pair kernel (k_f12_fold): 65,536 partial products took 16 launches,
// 2.4 ms -- 8 % of AggregateSignature::verify over 262,144 pairs.  fin and fout must not overlap (workgroups read what others write).
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_f12_tree_wide(size_t m, const uint32_t* fin, size_t sin, uint32_t* fout, size_t sout) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t base = (size_t)blockIdx.x * 16;
  if (base >= m) return;
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_F12_TREE16, WIDE_PROG_F12_TREE16_LEN);
  const int l = (int)(threadIdx.x & 15u);
  for (int u = (int)(threadIdx.x >> 4); u < 16 * 12; u += WIDE_TABLE_ROWS) {      // u = 12 j + v: value v of item j
    const int j = u / 12, v = u % 12, k = v >> 1, pw = k < 3 ? 2 * k : 2 * (k - 3) + 1;   // tower order -> powers of w, as k_finalexp_wide_ws
    uint32_t x = 0;
    if (base + j < m) {
      if (l < FP_NL) x = fin[(size_t)(W1 * v + l) * sin + base + j];
    } else if (v == 0 && l < FP_NL) {
      x = FP_ONE[l];
    }
    S.V[WV_L + 12 * j + 2 * pw + (v & 1)][l] = x;
  }
  __syncthreads();
  wide_exec(S, WIDE_PROG_F12_TREE16_LEN, K);
  for (int v = (int)(threadIdx.x >> 4); v < 12; v += WIDE_TABLE_ROWS) {
    const int k = v >> 1, pw = k < 3 ? 2 * k : 2 * (k - 3) + 1;
    if (l < FP_NL) fout[(size_t)(W1 * v + l) * sout + blockIdx.x] = S.V[WV_L + 2 * pw + (v & 1)][l];
  }
}
// ... the same for the fold trees of a pairing product's 68 entries (k_line_quad / k_f12_fold4), grid (ceil(qin / 16), 68): workgroup
// (b, e) <- the product of the values at positions e * rin + 16 b + j (j < 16, 16 b + j < qin), written at position e * rout + b of fout
// (a product on the engine takes 3.7 us against 43 on a lane pair: the levels that no longer fill the machine belong here)
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_f12_tree_seg(size_t qin, const uint32_t* fin, size_t sin, size_t rin, uint32_t* fout, size_t sout, size_t rout) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t first = (size_t)blockIdx.x * 16;
  if (first >= qin) return;
  const size_t base = (size_t)blockIdx.y * rin + first, have = qin - first;
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_F12_TREE16, WIDE_PROG_F12_TREE16_LEN);
  const int l = (int)(threadIdx.x & 15u);
  for (int u = (int)(threadIdx.x >> 4); u < 16 * 12; u += WIDE_TABLE_ROWS) {
    const int j = u / 12, v = u % 12, k = v >> 1, pw = k < 3 ? 2 * k : 2 * (k - 3) + 1;
    uint32_t x = 0;
    if ((size_t)j < have) {
      if (l < FP_NL) x = fin[(size_t)(W1 * v + l) * sin + base + j];
    } else if (v == 0 && l < FP_NL) {
      x = FP_ONE[l];
    }
    S.V[WV_L + 12 * j + 2 * pw + (v & 1)][l] = x;
  }
  __syncthreads();
  wide_exec(S, WIDE_PROG_F12_TREE16_LEN, K);
  for (int v = (int)(threadIdx.x >> 4); v < 12; v += WIDE_TABLE_ROWS) {
    const int k = v >> 1, pw = k < 3 ? 2 * k : 2 * (k - 3) + 1;
    if (l < FP_NL) fout[(size_t)(W1 * v + l) * sout + (size_t)blockIdx.y * rout + blockIdx.x] = S.V[WV_L + 2 * pw + (v & 1)][l];
  }
}
// ... and the Horner chain over the 68 per-entry products (items 0..67 of fin): program HORNER on one workgroup, 63 squarings and
// 67 products, the conjugated result (x < 0) as item 0 of fout -- the Miller product of all the items whose lines went in
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_f12_horner_wide(const uint32_t* fin, size_t sin, uint32_t* fout, size_t sout) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  if (blockIdx.x != 0) return;
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_HORNER, WIDE_PROG_HORNER_LEN);
  const int l = (int)(threadIdx.x & 15u);
  for (int u = (int)(threadIdx.x >> 4); u < WIDE_STEPS * 12; u += WIDE_TABLE_ROWS) {
    const int j = u / 12, v = u % 12, k = v >> 1, pw = k < 3 ? 2 * k : 2 * (k - 3) + 1;
    S.V[WV_L + 12 * j + 2 * pw + (v & 1)][l] = l < FP_NL ? fin[(size_t)(W1 * v + l) * sin + j] : 0u;
  }
  __syncthreads();
  wide_exec(S, WIDE_PROG_HORNER_LEN, K);
  for (int v = (int)(threadIdx.x >> 4); v < 12; v += WIDE_TABLE_ROWS) {
    const int k = v >> 1, pw = k < 3 ? 2 * k : 2 * (k - 3) + 1;
    if (l < FP_NL) fout[(size_t)(W1 * v + l) * sout] = S.V[WV_F + 2 * pw + (v & 1)][l];
  }
}
// The cut check, early parts.  blockIdx.y + first_part = 0: the key's line coefficients (program PRE_LINES: needs the key only);
// 1: the Miller function of the (signature, -g2) pair (PRE_F1: needs the signature only).  Both leave their result in the
// item's record; a single verification runs them side by side on two CUs while a third hashes the message.
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_pairing_pre(size_t n, uint32_t* rec, const int32_t* status, int first_part, int second_part) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t item = blockIdx.x;
  const int part = blockIdx.y == 0 ? first_part : second_part;
  if (item >= n) return;
  if (status[item] != BLS_OK) return;                   // uniform over the workgroup
  wide_consts K;
  wide_init(K);
  uint32_t* r = rec + item * WREC_WORDS;
  const int l = (int)(threadIdx.x & 15u), v = (int)(threadIdx.x >> 4);
  const uint32_t one_l = l < FP_NL ? FP_ONE[l] : 0u;
  if (part == 0) {
    wide_stage(S, WIDE_PROG_PRE_LINES, WIDE_PROG_PRE_LINES_LEN);
    if (v < 6) S.V[WV_PT0 + 6 + v][l] = r[16 * (WREC_Q0 + v) + l];      // Q, Jacobian (QPREP makes it homogeneous and sets T = Q)
    __syncthreads();
    wide_exec(S, WIDE_PROG_PRE_LINES_LEN, K);
    for (int t = threadIdx.x; t < WIDE_STEPS * 6 * 16; t += WIDE_ENGINE_BLOCK) {
      const int st = t / 96, w = t % 96;
      r[16 * WREC_L + 96 * st + w] = S.V[WV_L + 12 * st + (w >> 4)][w & 15];
    }
  } else {
    const int general = part == 2;                       // Bls12381G2Impl: pair 1 = (-g1, signature), lines from the signature
    if (general) wide_stage(S, WIDE_PROG_PRE_F1G, WIDE_PROG_PRE_F1G_LEN);
    else wide_stage(S, WIDE_PROG_PRE_F1, WIDE_PROG_PRE_F1_LEN);
    if (general) {
      if (v < 6) S.V[WV_PT1 + 6 + v][l] = r[16 * (WREC_Q1 + v) + l];
      if (v == 6) S.V[WV_P + 4][l] = l < FP_NL ? G1_GEN_X[l] : 0u;                               // P1 = -g1 = (x, -y, 1)
      if (v == 7) S.V[WV_P + 5][l] = l < FP_NL ? (uint32_t)(-(int32_t)G1_GEN_Y[l]) : 0u;
      if (v == 8) S.V[WV_P + 6][l] = one_l;
    } else {
      if (v < 3) S.V[WV_P + 4 + v][l] = r[16 * (WREC_P1 + v) + l];
      for (int t = threadIdx.x; t < WIDE_STEPS * 6 * 16; t += WIDE_ENGINE_BLOCK) {
        const int st = t / 96, w = (t % 96) >> 4, ll = t & 15;
        S.V[WV_L + 12 * st + 6 + w][ll] = ll < FP_NL ? G2NEGC_LINES[st][w * FP_NL + ll] : 0u;   // the lines of -[c] g2: P0 is uncleared
      }
    }
    if (v == 9) S.V[WV_P + 7][l] = 0u;
    if (threadIdx.x < 12 * 16) S.V[WV_F + v][l] = v == 0 ? one_l : 0u;          // f = 1
    __syncthreads();
    wide_exec(S, general ? WIDE_PROG_PRE_F1G_LEN : WIDE_PROG_PRE_F1_LEN, K);
    if (v < 12) r[16 * (WREC_F1 + v) + l] = S.V[WV_F + v][l];
  }
}
// The cut check, late part (program POST): the Miller function of (H(m), key) from the key's lines, times the other one, the
// final exponentiation and the comparison with one.
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_pairing_post(size_t n, const uint32_t* rec, int32_t* status) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t item = blockIdx.x;
  if (item >= n) return;
  if (status[item] != BLS_OK) return;                   // uniform over the workgroup
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_POST, WIDE_PROG_POST_LEN);
  const uint32_t* r = rec + item * WREC_WORDS;
  const int l = (int)(threadIdx.x & 15u), v = (int)(threadIdx.x >> 4);
  if (v < 3) S.V[WV_P + v][l] = r[16 * (WREC_P0 + v) + l];
  if (v == 3) S.V[WV_P + 3][l] = 0u;
  if (v >= 4) S.V[WV_F + v - 4][l] = (v == 4 && l < FP_NL) ? FP_ONE[l] : 0u;   // f = 1
  if (v < 12) S.V[WV_W + v][l] = r[16 * (WREC_F1 + v) + l];
  for (int t = threadIdx.x; t < WIDE_STEPS * 6 * 16; t += WIDE_ENGINE_BLOCK) {
    const int st = t / 96, w = t % 96;
    S.V[WV_L + 12 * st + (w >> 4)][w & 15] = r[16 * WREC_L + 96 * st + w];
  }
  if (threadIdx.x == 0) S.flag = 1;
  __syncthreads();
  wide_exec(S, WIDE_PROG_POST_LEN, K);
  if (threadIdx.x < 12) {
    fp x, one;
    w_load_local(x, S.V[WV_T + threadIdx.x]);
    bool ok;
    if (threadIdx.x == 0) {
      fp_one(one);
      ok = fp_eq(x, one);
    } else {
      ok = fp_is_zero(x);
    }
    if (!ok) S.flag = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) status[item] = S.flag ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}

// The streamed cut (declaration above).  Hand-over of a chunk (line steps [s0, s0 + ns), flag s0 / 4): the producer stores the chunk into the
// item's record, every thread fences, and after the workgroup's barrier thread 0 RELEASES flag = epoch at device scope; the
// consumer's thread 0 polls the flag with device-scope ACQUIRE loads, and after the barrier every thread fences and reads the
// chunk.  The two workgroups sit on different CUs -- as a rule on different XCDs with their own L2 -- so it is the device-scope
// release / acquire pair (L2 write-back on one side, invalidation on the other) that makes the lines visible, not the barrier.
// The poll is bounded: a consumer whose producer never arrives ends with BLS_ERR_STREAM_TIMEOUT instead of holding its CU.
static_assert((WIDE_STEPS + 3) / 4 + WREC_F2_SLOTS <= WSTREAM_FLAGS, "a flag per four line steps, the last ones for the Fp12 hand-over slots");
struct wide_stream_hook {
  wide_lds_t<wide_tb_f12>* S;
  uint32_t* r;
  uint32_t* flags;
  uint32_t epoch;
  // thread 0 waits for flag k (bounded), everybody learns the outcome and fences
  __device__ __forceinline__ bool wait_for(uint32_t k) const {
    if (threadIdx.x == 0) {
      int ok = 0;
      for (uint32_t spin = 0; spin < WSTREAM_SPIN_LIMIT; spin++) {
        if (__hip_atomic_load(flags + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
          ok = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      S->hook_ok = ok;
    }
    __syncthreads();
    if (!S->hook_ok) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
  }
  // every thread's stores so far become visible device-wide, then flag k says so
  __device__ __forceinline__ void announce(uint32_t k) const {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + k, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ bool operator()(uint32_t kind, uint32_t first, uint32_t count) const {
    if (kind == WOP_PUBF - WOP_FPINV || kind == WOP_ACQF - WOP_FPINV) {       // an Fp12 value: array `first` of the value store <-> hand-over slot `count`
      const int l = (int)(threadIdx.x & 15u), v = (int)(threadIdx.x >> 4);
      const uint32_t slot = count < WREC_F2_SLOTS ? count : 0u, at = 16 * (WREC_F2 + 12 * slot), k = WSTREAM_FLAGS - 1 - slot;
      if (kind == WOP_PUBF - WOP_FPINV) {
        if (v < 12) r[at + 16 * v + l] = S->V[first + v][l];
        announce(k);
        return true;
      }
      if (!wait_for(k)) return false;
      if (v < 12) S->V[first + v][l] = r[at + 16 * v + l];
      return true;
    }
    const int s0 = (int)first, ns = (int)count;
    const uint32_t k = first >> 2;
    if (kind == WOP_PUB - WOP_FPINV) {
      for (int t = threadIdx.x; t < ns * 96; t += WIDE_ENGINE_BLOCK) {
        const int st = s0 + t / 96, w = t % 96;
        r[16 * WREC_L + 96 * st + w] = S->V[WV_L + 12 * st + (w >> 4)][w & 15];
      }
      announce(k);
      return true;
    }
    if (!wait_for(k)) return false;
    for (int t = threadIdx.x; t < ns * 96; t += WIDE_ENGINE_BLOCK) {
      const int st = s0 + t / 96, w = t % 96;
      S->V[WV_L + 12 * st + (w >> 4)][w & 15] = r[16 * WREC_L + 96 * st + w];
    }
    return true;
  }
};
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_pairing_stream(size_t n, uint32_t* rec, int32_t* status, uint32_t* flags, uint32_t epoch) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t item = blockIdx.x;
  if (item >= n) return;
  if (status[item] != BLS_OK) return;                   // uniform over the workgroup, and the same in all workgroups of the item:
  wide_consts K;                                        // the last consumer writes the status only after everybody's hand-overs
  wide_init(K);
  uint32_t* r = rec + item * WREC_WORDS;
  const wide_stream_hook hook = {&S, r, flags + item * WSTREAM_FLAGS, epoch};
  const int l = (int)(threadIdx.x & 15u), v = (int)(threadIdx.x >> 4);
  if (blockIdx.y == 0) {                                // the producer: pair 0's lines
    wide_stage(S, WIDE_PROG_PRE_LINES_S, WIDE_PROG_PRE_LINES_S_LEN);
    if (v < 6) S.V[WV_PT0 + 6 + v][l] = r[16 * (WREC_Q0 + v) + l];      // Q, Jacobian (QPREP makes it homogeneous and sets T = Q)
    __syncthreads();
    wide_exec(S, WIDE_PROG_PRE_LINES_S_LEN, K, hook);
    return;
  }
  const bool lo = blockIdx.y == 1;                      // the Miller loop's last iterations from 1 (POST_LO_S); the other one: the rest (POST_HI_S)
  if (lo) wide_stage(S, WIDE_PROG_POST_LO_S, WIDE_PROG_POST_LO_S_LEN);
  else wide_stage(S, WIDE_PROG_POST_HI_S, WIDE_PROG_POST_HI_S_LEN);
  if (v < 3) S.V[WV_P + v][l] = r[16 * (WREC_P0 + v) + l];
  if (v == 3) S.V[WV_P + 3][l] = 0u;
  if (v >= 4) S.V[WV_F + v - 4][l] = (v == 4 && l < FP_NL) ? FP_ONE[l] : 0u;   // f = 1
  if (!lo && v < 12) S.V[WV_W + v][l] = r[16 * (WREC_F1 + v) + l];
  if (threadIdx.x == 0) S.flag = 1;
  __syncthreads();
  if (lo) {
    wide_exec(S, WIDE_PROG_POST_LO_S_LEN, K, hook);     // a timeout here leaves the partner without its value: it times out in turn
    return;
  }
  if (!wide_exec(S, WIDE_PROG_POST_HI_S_LEN, K, hook)) {
    if (threadIdx.x == 0) status[item] = BLS_ERR_STREAM_TIMEOUT;
    return;
  }
  if (threadIdx.x < 12) {
    fp x, one;
    w_load_local(x, S.V[WV_T + threadIdx.x]);
    bool ok;
    if (threadIdx.x == 0) {
      fp_one(one);
      ok = fp_eq(x, one);
    } else {
      ok = fp_is_zero(x);
    }
    if (!ok) S.flag = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) status[item] = S.flag ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_pairing_post2(size_t n, uint32_t* rec, int32_t* status, uint32_t* flags, uint32_t epoch) {
  __shared__ wide_lds_t<wide_tb_f12> S;
  const size_t item = blockIdx.x;
  if (item >= n) return;
  if (status[item] != BLS_OK) return;                   // as in k_pairing_stream: the same in all workgroups of the item
  wide_consts K;
  wide_init(K);
  uint32_t* r = rec + item * WREC_WORDS;
  const wide_stream_hook hook = {&S, r, flags + item * WSTREAM_FLAGS, epoch};
  // gridDim.y = 2 or 3 workgroups per item; the LAST block index is part 0 (it waits for the others, which are placed first)
  const int part = (int)(gridDim.y - 1 - blockIdx.y);
  const uint32_t* prog;
  int len;
  if (gridDim.y == 3) {
    prog = part == 0 ? WIDE_PROG_POST3_HI : (part == 1 ? WIDE_PROG_POST3_MID : WIDE_PROG_POST3_LO);
    len = part == 0 ? WIDE_PROG_POST3_HI_LEN : (part == 1 ? WIDE_PROG_POST3_MID_LEN : WIDE_PROG_POST3_LO_LEN);
  } else {
    prog = part == 0 ? WIDE_PROG_POST_HI : WIDE_PROG_POST_LO;
    len = part == 0 ? WIDE_PROG_POST_HI_LEN : WIDE_PROG_POST_LO_LEN;
  }
  wide_stage(S, prog, len);
  const int l = (int)(threadIdx.x & 15u), v = (int)(threadIdx.x >> 4);
  if (v < 3) S.V[WV_P + v][l] = r[16 * (WREC_P0 + v) + l];
  if (v == 3) S.V[WV_P + 3][l] = 0u;
  if (v >= 4) S.V[WV_F + v - 4][l] = (v == 4 && l < FP_NL) ? FP_ONE[l] : 0u;   // f = 1
  if (part == 0 && v < 12) S.V[WV_W + v][l] = r[16 * (WREC_F1 + v) + l];
  for (int t = threadIdx.x; t < WIDE_STEPS * 6 * 16; t += WIDE_ENGINE_BLOCK) {
    const int st = t / 96, w = t % 96;
    S.V[WV_L + 12 * st + (w >> 4)][w & 15] = r[16 * WREC_L + 96 * st + w];
  }
  if (threadIdx.x == 0) S.flag = 1;
  __syncthreads();
  if (part != 0) {
    wide_exec(S, len, K, hook);
    return;
  }
  if (!wide_exec(S, len, K, hook)) {
    if (threadIdx.x == 0) status[item] = BLS_ERR_STREAM_TIMEOUT;
    return;
  }
  if (threadIdx.x < 12) {
    fp x, one;
    w_load_local(x, S.V[WV_T + threadIdx.x]);
    bool ok;
    if (threadIdx.x == 0) {
      fp_one(one);
      ok = fp_eq(x, one);
    } else {
      ok = fp_is_zero(x);
    }
    if (!ok) S.flag = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) status[item] = S.flag ? BLS_OK : BLS_ERR_INVALID_SIGNATURE;
}

// The tail of every point sum (MultiPublicKey::from_public_keys, the windows of a multi-scalar multiplication): a fold level of
// the lane-local / lane-pair kernels is one addition deep and costs ~50 us however few points are left, sixteen levels for a
// million keys.  Here one workgroup sums SIXTEEN points -- four levels of complete projective additions as table operations
// (csrc/wide_tables.cuh, set PT; ~35 us) -- so 4,096 partial sums are one point after three launches.  in / out: RAW_PROJ
// (Jacobian; an identity comes back as Z = 0).
template <int G>
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_point_tree_wide(size_t m, const uint8_t* in, uint8_t* out) {
  __shared__ wide_lds_t<wide_tb_pt> S;
  constexpr int NV = G == 1 ? 3 : 6, PSZ = G == 1 ? 144 : 288;          // Fp values and bytes per point
  const size_t base = (size_t)blockIdx.x * WIDE_PT_POINTS;
  wide_consts K;
  wide_init(K);
  wide_stage(S, G == 1 ? WIDE_PROG_G1_JJ : WIDE_PROG_G2_JJ, G == 1 ? WIDE_PROG_G1_JJ_LEN : WIDE_PROG_G2_JJ_LEN);
  if (threadIdx.x < WIDE_PT_POINTS * NV) {
    const int i = (int)threadIdx.x / NV, v = (int)threadIdx.x % NV;
    const int slot = G == 1 ? WIDE_PT_SLOT_G1 * (i >> 1) + 3 * (i & 1) : WIDE_PT_SLOT_G2 * (i >> 1) + 6 * (i & 1);
    fp x;
    bool inf = base + i >= m;
    if (!inf) {
      const uint32_t* w = (const uint32_t*)(in + (base + i) * PSZ);
      inf = words_all_zero(w + (G == 1 ? 24 : 48), G == 1 ? 12 : 24);     // Z = 0
      if (!inf) fp_from_raw(x, w + 12 * v);
    }
    if (inf) {                                                             // the identity as (0, 1, 0)
      if (v == (G == 1 ? 1 : 2)) fp_one(x);
      else fp_zero(x);
    }
    fp_reduce(x, x);
    w_store_local(S.V[WPV_L0 + slot + v], x);
  }
  __syncthreads();
  wide_exec(S, G == 1 ? WIDE_PROG_G1_JJ_LEN : WIDE_PROG_G2_JJ_LEN, K);
  if (threadIdx.x < NV) {
    fp x;
    w_load_local(x, S.V[WPV_L4 + threadIdx.x]);
    fp_to_raw((uint32_t*)(out + (size_t)blockIdx.x * PSZ) + 12 * threadIdx.x, x);
  }
}
template __global__ void k_point_tree_wide<1>(size_t, const uint8_t*, uint8_t*);
template __global__ void k_point_tree_wide<2>(size_t, const uint8_t*, uint8_t*);

// hash-to-G2 of a few messages spends more than half its time in the cofactor clearing when a lane pair walks it (two scalar
// multiplications by |x|: 126 doublings and 20 additions of ~20 us each).  Here it is program G2_CLEAR of table set PT: 142
// complete projective additions as two table steps each, psi as one (conjugates are linear, its two constants sit in the
// value store), ~0.45 ms.  In place on RAW_PROJ points.
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_g2_clear_wide(size_t n, uint8_t* pts) {
  __shared__ wide_lds_t<wide_tb_pt> S;
  const size_t i = blockIdx.x;
  if (i >= n) return;
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_G2_CLEAR, WIDE_PROG_G2_CLEAR_LEN);
  uint32_t* w = (uint32_t*)(pts + i * 288);
  if (threadIdx.x < 6) {
    const int v = (int)threadIdx.x;
    fp x;
    if (words_all_zero(w + 48, 24)) {            // Z = 0: the identity as (0, 1, 0)
      if (v == 2) fp_one(x);
      else fp_zero(x);
    } else {
      fp_from_raw(x, w + 12 * v);
    }
    fp_reduce(x, x);
    w_store_local(S.V[WPV_R0 + v], x);
  }
  if (threadIdx.x >= 64 && threadIdx.x < 64 + 4 * 16) {   // psi's constants cx, cy (real, imaginary limbs)
    const int t = (int)threadIdx.x - 64, v = t >> 4, l = t & 15;
    const uint32_t* src = v < 2 ? PSI_CX : PSI_CY;
    S.V[WPV_CONST + v][l] = l < FP_NL ? src[(v & 1) * FP_NL + l] : 0u;
  }
  __syncthreads();
  wide_exec(S, WIDE_PROG_G2_CLEAR_LEN, K);
  if (threadIdx.x < 6) {
    fp x;
    w_load_local(x, S.V[WPV_R3 + threadIdx.x]);
    fp_to_raw(w + 12 * threadIdx.x, x);
  }
}

// ---- hash_to_curve to G1 for single items: ONE wave per message, the two SSWU maps on rows 0 and 1 of the wave in the
// row-wide field type `wf` (csrc/wide.cuh), then both rows add the two points and clear the cofactor redundantly.
// A dependent multiplication costs ~0.45 us here against ~1.2 us lane-local, and this hash is ~1,000 of them in a row
// (two square-root exponentiations side by side, the 11-isogeny, 64 doublings).  out: RAW_PROJ points.
__device__ __forceinline__ void iso1_poly_wide(wf& r, const uint32_t (*k)[FP_NL], int deg, const wf& xn) {
  const uint32_t (*zp)[16] = g_wf.tab[wf_row()];           // zp[j] = xd^j, staged by the caller
  const int l = wf_lane();
  wf acc, c, t, z;
  fp_load(acc, k[deg]);
  for (int i = deg - 1; i >= 0; i--) {
    fp_mul(acc, acc, xn);
    fp_load(c, k[i]);
    z.v = (wfp)zp[deg - i][l];
    fp_mul(t, c, z);
    fp_add(acc, acc, t);
  }
  r = acc;
}
__device__ __noinline__ void iso_map_g1_wide(jac<wf>& r, const wf& xn, const wf& xd, const wf& y) {
  uint32_t (*zp)[16] = g_wf.tab[wf_row()];
  const int l = wf_lane();
  wf t, XN, XD, YN, YD, zx, yd2;
  fp_one(t);
  zp[0][l] = (uint32_t)t.v;
  zp[1][l] = (uint32_t)xd.v;
  t = xd;
  for (int i = 2; i < 16; i++) {
    fp_mul(t, t, xd);
    zp[i][l] = (uint32_t)t.v;
  }
  iso1_poly_wide(XN, ISO1_XNUM, 11, xn);
  iso1_poly_wide(XD, ISO1_XDEN, 10, xn);
  iso1_poly_wide(YN, ISO1_YNUM, 15, xn);
  iso1_poly_wide(YD, ISO1_YDEN, 15, xn);
  fp_mul(zx, XD, xd);     // x_out = XN / zx,  y_out = y YN / YD
  fp_mul(r.z, zx, YD);    // Z = zx YD
  fp_sqr(yd2, YD);
  fp_mul(t, XN, zx);
  fp_mul(r.x, t, yd2);    // X = XN zx YD^2
  fp_sqr(t, zx);
  fp_mul(t, t, zx);
  fp_mul(t, t, yd2);
  fp_mul(t, t, YN);
  fp_mul(r.y, t, y);      // Y = y YN zx^3 YD^2
}
// One G1 doubling by TWO rows that hold the same point (the schedule of h2c.cuh's jac_dbl_pair one level up): the seven
// field products of dbl-2009-l as FOUR steps -- row 0: A = X^2, F = (3A)^2, YZ, E (D - X3); row 1: B = Y^2, C = B^2,
// (X + B)^2 -- the rows exchange single registers (an element is one VGPR per lane here).  Same operation order and
// reductions as jac_dbl_body.
__device__ __forceinline__ void wf_row_swap(wf& r, const wf& a) { r.v = __shfl_xor(a.v, 16, 64); }
__device__ __forceinline__ void wf_sel(wf& r, bool c, const wf& a, const wf& b) { r.v = c ? a.v : b.v; }
__device__ __forceinline__ void jac_dbl_rows(jac<wf>& p, bool hi) {
  wf s1, s2, s3, in, a, b, o2, o3, D, E, t, x3, y3, z3, c8;
  wf_sel(in, hi, p.y, p.x);
  fp_sqr(s1, in);                 // row 0: A = X^2           row 1: B = Y^2
  fp_dbl(E, s1);
  fp_add(E, E, s1);
  fp_reduce(E, E);                // row 0: E = 3A
  wf_sel(in, hi, s1, E);
  fp_sqr(s2, in);                 // row 0: F = E^2           row 1: C = B^2
  fp_add(t, p.x, s1);             //                          row 1: X + B
  wf_sel(a, hi, t, p.y);
  wf_sel(b, hi, t, p.z);
  fp_mul(s3, a, b);               // row 0: Y Z               row 1: (X + B)^2
  wf_row_swap(o2, s2);            // row 0: C
  wf_row_swap(o3, s3);            // row 0: (X + B)^2
  fp_sub(t, o3, s1);
  fp_sub(t, t, o2);
  fp_dbl(D, t);
  fp_reduce(D, D);                // D = 2((X + B)^2 - A - C)
  fp_dbl(t, D);
  fp_sub(t, s2, t);
  fp_reduce(x3, t);               // X3 = F - 2D
  fp_sub(t, D, x3);
  fp_mul(t, E, t);                // E (D - X3)
  fp_dbl(c8, o2);
  fp_dbl(c8, c8);
  fp_reduce(c8, c8);
  fp_dbl(c8, c8);                 // 8C
  fp_sub(t, t, c8);
  fp_reduce(y3, t);
  fp_dbl(z3, s3);
  fp_reduce(z3, z3);              // Z3 = 2YZ
  wf_row_swap(a, x3);
  wf_row_swap(b, y3);
  wf_row_swap(t, z3);
  wf_sel(p.x, hi, a, x3);
  wf_sel(p.y, hi, b, y3);
  wf_sel(p.z, hi, t, z3);
}
// [k] P (left to right), the doublings shared by the two rows; both rows hold the same P and end with the same result
__device__ __noinline__ void jac_mul_u64_rows(jac<wf>& r, const jac<wf>& p, uint64_t k, bool hi) {
  jac<wf> acc;
  jac_set_inf(acc);
  for (int i = 63; i >= 0; i--) {
    jac_dbl_rows(acc, hi);                        // the identity (Z = 0) stays the identity: Z3 = 2 Y Z
    if ((k >> i) & 1) jac_add(acc, acc, p);
  }
  r = acc;
}
// ---- expand_message_xmd (RFC 9380 5.3.1, SHA-256) by a whole WAVE for one message.  The streaming one-lane version of
// h2c.cuh appends byte by byte into a word array with a run-time index, i.e. a scratch read-modify-write per byte: 0.36 ms
// of pure latency for the ~300 bytes of a 32-byte message -- a third of the single-message hash.  Here lane L fetches byte L of
// the current 64-byte block straight from where it lives (zero pad, message, length / DST suffix, padding), the block meets in
// LDS, and every lane runs the compression on registers (fully unrolled, the sixteen message words passed by value): ~5 us per
// block, ~11 blocks.  All 64 lanes of the wave must call it; every lane returns the same words.
typedef u32x8_t u32x8;      // (h2c.cuh: the compression with its operands by value, sha256_compress_v)
typedef u32x16_t u32x16;
__device__ __forceinline__ u32x8 sha256_iv() {
  u32x8 h;
  h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
  h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
  return h;
}
// the block that the lanes wrote byte by byte (stream order) -> sixteen big-endian words on every lane
__device__ __forceinline__ u32x16 sha256_block_from_lds(uint8_t* blk) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  u32x16 w;
  const uint32_t* q = (const uint32_t*)blk;
#pragma unroll
  for (int j = 0; j < 16; j++) w[j] = __builtin_bswap32(q[j]);
  __builtin_amdgcn_wave_barrier();
  return w;
}
// out: NOUT / 4 big-endian words (word 0 = the first four bytes of the uniform bytes).  blk: 64 bytes of LDS owned by this wave.
template <int NOUT>
__device__ __forceinline__ void expand_message_xmd_wave(uint32_t* out, const uint8_t* m, uint32_t m_len, const dst_arg& dst, uint8_t* blk) {
  const uint32_t L = threadIdx.x & 63u;
  const uint32_t dl = dst.len;
  // b0 = H(Z_pad || msg || I2OSP(NOUT, 2) || 0 || DST || len(DST)): the stream after the 64 zero bytes, whose compression from
  // the initial state is data-independent but is simply run (one block of eleven)
  const uint32_t total = 64 + m_len + 3 + dl + 1;
  const uint32_t nblk = (total + 9 + 63) >> 6;
  u32x8 h = sha256_iv();
  for (uint32_t kb = 0; kb < nblk; kb++) {
    uint32_t pos = kb * 64 + L, byte = 0;
    if (pos >= 64 && pos < total) {
      uint32_t q = pos - 64;
      if (q < m_len) {
        byte = m[q];
      } else {
        q -= m_len;
        if (q == 0) byte = (uint32_t)(NOUT >> 8);
        else if (q == 1) byte = (uint32_t)(NOUT & 255);
        else if (q == 2) byte = 0;
        else if (q - 3 < dl) byte = dst.b[q - 3];
        else byte = dl;
      }
    } else if (pos == total) {
      byte = 0x80;
    } else if (kb == nblk - 1 && L >= 60) {
      byte = ((total << 3) >> (8 * (63 - L))) & 255u;          // the bit length, big-endian (below 2^32)
    }
    blk[L] = (uint8_t)byte;
    h = sha256_compress_v(h, sha256_block_from_lds(blk));
  }
  const u32x8 b0 = h;
  u32x8 prev;
#pragma unroll
  for (int j = 0; j < 8; j++) prev[j] = 0;
  const uint32_t total_i = 32 + 1 + dl + 1, nblk_i = (total_i + 9 + 63) >> 6;
  for (int bi = 1; bi <= NOUT / 32; bi++) {
    u32x8 x;
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = b0[j] ^ prev[j];
    h = sha256_iv();
    for (uint32_t kb = 0; kb < nblk_i; kb++) {
      const uint32_t pos = kb * 64 + L;
      uint32_t byte = 0;
      if (pos < 32) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) word = (pos >> 2) == (uint32_t)j ? x[j] : word;
        byte = (word >> (8 * (3 - (pos & 3)))) & 255u;
      } else if (pos == 32) {
        byte = (uint32_t)bi;
      } else if (pos < 33 + dl) {
        byte = dst.b[pos - 33];
      } else if (pos == 33 + dl) {
        byte = dl;
      } else if (pos == total_i) {
        byte = 0x80;
      } else if (kb == nblk_i - 1 && L >= 60) {
        byte = ((total_i << 3) >> (8 * (63 - L))) & 255u;
      }
      blk[L] = (uint8_t)byte;
      h = sha256_compress_v(h, sha256_block_from_lds(blk));
    }
    prev = h;
#pragma unroll
    for (int j = 0; j < 8; j++) out[8 * (bi - 1) + j] = h[j];
  }
}
// 64 uniform bytes given as sixteen big-endian words -> Fp (Montgomery), as fp_from_be64
__device__ __forceinline__ void fp_from_be_words(fp& r, const uint32_t* bw) {
  fp lo, hi, t, k;
  uint32_t w[12];
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = i < 4 ? bw[3 - i] : 0u;
  fp_set_words(hi, w);
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = bw[15 - i];
  fp_set_words(lo, w);
  fp_load(k, FP_R2);
  fp_mul(t, k, lo);
  fp_load(k, FP_R2_384);
  fp_mul(hi, k, hi);
  fp_add(t, t, hi);
  fp_reduce(r, t);
}

__global__ void __launch_bounds__(WIDE_BLOCK) k_hash_to_g1_wide(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst, uint8_t* out,
                                                             uint32_t* rec) {
  __shared__ uint32_t pts[4][2][3][16];     // per wave: the Jacobian points of the two maps
  __shared__ __attribute__((aligned(16))) uint8_t shablk[4][64];
  const int wave = threadIdx.x >> 6, row = (threadIdx.x >> 4) & 3, l = threadIdx.x & 15;
  const size_t i = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
  wf_setup();
  __syncthreads();
  if (i >= n) return;
  const int stop = single_msg >> 8;            // measurement aid (blsgpu.hip hash_phase_stop): leave after phase `stop`
  const size_t mi = (single_msg & 1) ? 0 : i;
  uint32_t ubw[32];
  expand_message_xmd_wave<128>(ubw, msgs + offs[mi], (uint32_t)(offs[mi + 1] - offs[mi]), dst, shablk[wave]);   // the whole wave
  if (row >= 2) return;                       // rows 2 and 3 of the wave stay idle from here: the hash has two independent maps
  uint32_t uw[16];
#pragma unroll
  for (int j = 0; j < 16; j++) uw[j] = row == 1 ? ubw[16 + j] : ubw[j];    // u0 on row 0, u1 on row 1
  fp ul;
  fp_from_be_words(ul, uw);                   // every lane of the row computes the same value
  wf u, xn, xd, y;
  wf_from_local(u, ul);
  if (stop == 1) return;
  sswu_g1(xn, xd, y, u);
  if (stop == 2) return;
  jac<wf> mine, q0, q1, acc;
  iso_map_g1_wide(mine, xn, xd, y);
  if (stop == 3) return;
  // both rows take both points and continue with identical operands
  pts[wave][row][0][l] = (uint32_t)mine.x.v;
  pts[wave][row][1][l] = (uint32_t)mine.y.v;
  pts[wave][row][2][l] = (uint32_t)mine.z.v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  q0.x.v = (wfp)pts[wave][0][0][l];
  q0.y.v = (wfp)pts[wave][0][1][l];
  q0.z.v = (wfp)pts[wave][0][2][l];
  q1.x.v = (wfp)pts[wave][1][0][l];
  q1.y.v = (wfp)pts[wave][1][1][l];
  q1.z.v = (wfp)pts[wave][1][2][l];
  jac_add(q0, q0, q1);
  if (stop == 4) return;
  if ((rec != nullptr && out == nullptr) || (single_msg & 2)) {   // the record of a cut check takes the point uncleared (see k_hash_to_g1_engine); bit 1 of single_msg asks for it in `out` as well
    acc = q0;
  } else {
    jac_mul_u64_rows(q1, q0, BLS_X_ABS, row == 1);   // clear cofactor: h_eff = 1 - x = 1 + |x|
    if (stop == 5) return;
    jac_add(acc, q1, q0);
  }
  if (row == 0 && rec) {                       // the cut check takes H(m) as engine values, Jacobian, straight from the row
    uint32_t* r = rec + i * WREC_WORDS + 16 * WREC_P0;
    wf t;
    fp_reduce(t, acc.x);
    r[l] = (uint32_t)t.v;
    fp_reduce(t, acc.y);
    r[16 + l] = (uint32_t)t.v;
    fp_reduce(t, acc.z);
    r[32 + l] = (uint32_t)t.v;
  }
  if (row == 0 && out) {
    fp X, Y, Z;
    wf_to_local(X, acc.x);
    wf_to_local(Y, acc.y);
    wf_to_local(Z, acc.z);
    if (l == 0) {
      g1_jac h;
      h.x = X;
      h.y = Y;
      h.z = Z;
      store_g1_pt(out, i, h);
    }
  }
}
// One workgroup per message: wave 0 runs expand_message_xmd and the two SSWU maps (the square-root chains: dependent
// multiplications, the row-wide field type's business); everything behind them is program G1_HASH_TAIL of table set PT on all
// four waves -- the 11-isogeny of both points as eight table steps (every polynomial ONE linear row over monomial x constant
// products: 15 us where the Horner chains took 87), their sum, and the cofactor clearing (1 + |x|) q as 70 complete additions
// (0.17 ms where the two rows took 0.29).  (A point on a pole of the isogeny would need the identity here; no message hashes there.)
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_hash_to_g1_engine(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst,
                                                                       uint8_t* out, uint32_t* rec) {
  __shared__ wide_lds_t<wide_tb_pt> S;
  __shared__ __attribute__((aligned(16))) uint8_t shablk[64];
  const int wave = threadIdx.x >> 6, row = (threadIdx.x >> 4) & 3, l = threadIdx.x & 15;
  const size_t i = blockIdx.x;
  if (i >= n) return;
  wide_consts K;
  wide_init(K);
  wf_setup();
  // the record of a cut check takes the message point UNCLEARED (its second pair is (sig, -[c] g2): csrc/g2neg_lines.cuh)
  const bool no_clear = rec != nullptr && out == nullptr;
  const int prog_len = no_clear ? WIDE_PROG_G1_HASH_MAP_LEN : WIDE_PROG_G1_HASH_TAIL_LEN, res = no_clear ? WPV_R0 : WPV_R1;
  wide_stage(S, no_clear ? WIDE_PROG_G1_HASH_MAP : WIDE_PROG_G1_HASH_TAIL, prog_len);
  __syncthreads();
  if (wave == 0) {
    const size_t mi = (single_msg & 1) ? 0 : i;
    uint32_t ubw[32];
    expand_message_xmd_wave<128>(ubw, msgs + offs[mi], (uint32_t)(offs[mi + 1] - offs[mi]), dst, shablk);   // the whole wave
    if (row < 2) {
      uint32_t uw[16];
#pragma unroll
      for (int j = 0; j < 16; j++) uw[j] = row == 1 ? ubw[16 + j] : ubw[j];    // u0 on row 0, u1 on row 1
      fp ul;
      fp_from_be_words(ul, uw);
      wf u, xn, xd, y, t;
      wf_from_local(u, ul);
      sswu_g1(xn, xd, y, u);
      uint32_t (*iso)[16] = &S.V[WPV_ISO + WIDE_PT_ISO_STRIDE * row];      // x' = xn / xd and y of map `row`
      fp_reduce(t, xn);
      iso[0][l] = (uint32_t)t.v;
      fp_reduce(t, xd);
      iso[1][l] = (uint32_t)t.v;
      fp_reduce(t, y);
      iso[2][l] = (uint32_t)t.v;
    }
  } else {                                                  // meanwhile: the isogeny's 55 coefficients into the constants (xnum, xden, ynum, yden)
    for (int t = (int)threadIdx.x - 64; t < 55 * 16; t += WIDE_ENGINE_BLOCK - 64) {
      const int v = t >> 4, ll = t & 15;
      const uint32_t* src = v < 12 ? ISO1_XNUM[v] : (v < 23 ? ISO1_XDEN[v - 12] : (v < 39 ? ISO1_YNUM[v - 23] : ISO1_YDEN[v - 39]));
      S.V[WPV_CONST + WIDE_PT_ISO_K + v][ll] = ll < FP_NL ? src[ll] : 0u;
    }
  }
  __syncthreads();
  wide_exec(S, prog_len, K);
  const int v = (int)(threadIdx.x >> 4);
  if (rec && v < 3) rec[i * WREC_WORDS + 16 * (WREC_P0 + v) + l] = S.V[res + v][l];
  if (out && threadIdx.x < 3) {
    fp x;
    w_load_local(x, S.V[res + threadIdx.x]);
    fp_to_raw((uint32_t*)(out + i * 144) + 12 * threadIdx.x, x);
  }
}
// one SSWU map and its 3-isogeny on this DPP row (row 0: u0, row 1: u1 of the 64 expanded words): the point of E2, homogeneous,
// as six engine values at pt (LDS or global)
__device__ __forceinline__ void hash_g2_map_row(const uint32_t* ubw, int row, int l, uint32_t (*pt)[16]) {
  uint32_t uw[16];
  fp ul;
  wf2 u, x, y, X, Y, Z, one;
#pragma unroll
  for (int j = 0; j < 16; j++) uw[j] = row == 1 ? ubw[32 + j] : ubw[j];
  fp_from_be_words(ul, uw);
  wf_from_local(u.c0, ul);
#pragma unroll
  for (int j = 0; j < 16; j++) uw[j] = row == 1 ? ubw[48 + j] : ubw[16 + j];
  fp_from_be_words(ul, uw);
  wf_from_local(u.c1, ul);
  sswu_g2(x, y, u);
  iso_map_g2_hom(X, Y, Z, x, y);
  const bool inf = fp2_is_zero(Z);                      // an exceptional point of the isogeny: the identity (0 : 1 : 0)
  fp2_one(one);
  fp2_cmov(Y, one, inf);
  fp_zero(one.c0);
  fp2_cmov(X, one, inf);
  fp2_reduce(X, X);
  fp2_reduce(Y, Y);
  fp2_reduce(Z, Z);
  pt[0][l] = (uint32_t)X.c0.v;
  pt[1][l] = (uint32_t)X.c1.v;
  pt[2][l] = (uint32_t)Y.c0.v;
  pt[3][l] = (uint32_t)Y.c1.v;
  pt[4][l] = (uint32_t)Z.c0.v;
  pt[5][l] = (uint32_t)Z.c1.v;
}
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_hash_to_g2_engine(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst,
                                                                       uint8_t* out, uint32_t* rec) {
  __shared__ wide_lds_t<wide_tb_pt> S;
  __shared__ __attribute__((aligned(16))) uint8_t shablk[64];
  const int wave = threadIdx.x >> 6, row = (threadIdx.x >> 4) & 3, l = threadIdx.x & 15;
  const size_t i = blockIdx.x;
  if (i >= n) return;
  wide_consts K;
  wide_init(K);
  wf_setup();
  wide_stage(S, WIDE_PROG_G2_HASH_TAIL, WIDE_PROG_G2_HASH_TAIL_LEN);
  __syncthreads();
  if (wave == 0) {
    const size_t mi = (single_msg & 1) ? 0 : i;
    uint32_t ubw[64];
    expand_message_xmd_wave<256>(ubw, msgs + offs[mi], (uint32_t)(offs[mi + 1] - offs[mi]), dst, shablk);   // the whole wave
    if (row < 2) hash_g2_map_row(ubw, row, l, &S.V[row == 1 ? WPV_R1 : WPV_R0]);
  } else if (threadIdx.x < 64 + 4 * 16) {                   // meanwhile: psi's constants cx, cy (real, imaginary limbs)
    const int t = (int)threadIdx.x - 64, v = t >> 4;
    const uint32_t* src = v < 2 ? PSI_CX : PSI_CY;
    S.V[WPV_CONST + v][l] = l < FP_NL ? src[(v & 1) * FP_NL + l] : 0u;
  }
  __syncthreads();
  wide_exec(S, WIDE_PROG_G2_HASH_TAIL_LEN, K);
  if (rec && (threadIdx.x >> 4) < 6)     // the cut check's record takes H(m) as it stands here: six engine values, Jacobian (what k_prepare_keys part 4 would write)
    rec[i * WREC_WORDS + 16 * (WREC_Q0 + (threadIdx.x >> 4)) + (threadIdx.x & 15u)] = S.V[WPV_R3 + (threadIdx.x >> 4)][threadIdx.x & 15u];
  if (out && threadIdx.x < 6) {
    fp x;
    w_load_local(x, S.V[WPV_R3 + threadIdx.x]);
    fp_to_raw((uint32_t*)(out + i * 288) + 12 * threadIdx.x, x);
  }
}
// The same in two launches for more messages than there are CUs: one WAVE per message for the maps (pts: twelve engine values
// per message, the two points), then one workgroup per message for the engine program.
__global__ void __launch_bounds__(WIDE_BLOCK) k_hash_to_g2_maps(size_t n, const uint8_t* msgs, const uint64_t* offs, int single_msg, dst_arg dst, uint32_t* pts) {
  __shared__ __attribute__((aligned(16))) uint8_t shablk[4][64];
  const int wave = threadIdx.x >> 6, row = (threadIdx.x >> 4) & 3, l = threadIdx.x & 15;
  const size_t i = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
  wf_setup();
  __syncthreads();
  if (i >= n) return;
  const size_t mi = (single_msg & 1) ? 0 : i;
  uint32_t ubw[64];
  expand_message_xmd_wave<256>(ubw, msgs + offs[mi], (uint32_t)(offs[mi + 1] - offs[mi]), dst, shablk[wave]);
  if (row < 2) hash_g2_map_row(ubw, row, l, (uint32_t (*)[16])(pts + (i * 12 + 6 * row) * 16));
}
__global__ void __launch_bounds__(WIDE_ENGINE_BLOCK) k_g2_hash_tail_wide(size_t n, const uint32_t* pts, uint8_t* out, uint32_t* rec) {
  __shared__ wide_lds_t<wide_tb_pt> S;
  const size_t i = blockIdx.x;
  if (i >= n) return;
  wide_consts K;
  wide_init(K);
  wide_stage(S, WIDE_PROG_G2_HASH_TAIL, WIDE_PROG_G2_HASH_TAIL_LEN);
  const int v = (int)(threadIdx.x >> 4), l = (int)(threadIdx.x & 15u);
  if (v < 12) S.V[(v < 6 ? WPV_R0 : WPV_R1 - 6) + v][l] = pts[(i * 12 + v) * 16 + l];
  else if (v < 16) {
    const uint32_t* src = v < 14 ? PSI_CX : PSI_CY;
    S.V[WPV_CONST + v - 12][l] = l < FP_NL ? src[(v & 1) * FP_NL + l] : 0u;
  }
  __syncthreads();
  wide_exec(S, WIDE_PROG_G2_HASH_TAIL_LEN, K);
  if (rec && (threadIdx.x >> 4) < 6)     // the cut check's record takes H(m) as it stands here: six engine values, Jacobian (what k_prepare_keys part 4 would write)
    rec[i * WREC_WORDS + 16 * (WREC_Q0 + (threadIdx.x >> 4)) + (threadIdx.x & 15u)] = S.V[WPV_R3 + (threadIdx.x >> 4)][threadIdx.x & 15u];
  if (out && threadIdx.x < 6) {
    fp x;
    w_load_local(x, S.V[WPV_R3 + threadIdx.x]);
    fp_to_raw((uint32_t*)(out + i * 288) + 12 * threadIdx.x, x);
  }
}
#endif  // BLS_TU_WIDE
