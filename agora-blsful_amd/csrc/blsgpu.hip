// libblsgpu.so: host side of the C ABI declared in include/blsgpu.h.  HIP runtime only (no torch types).
// There is no CPU compute path in this library: without a gfx950 device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/blsgpu.h"
#include "kernels.cuh"
#include "host_sha256.h"

namespace {

const uint32_t FP_ONE_HOST[12] = FP_RAW_ONE_WORDS;   // 1 as the C ABI's Fp12 records carry it
thread_local std::string t_err;

enum {
  KID_PREPARE, KID_MILLER2, KID_FINALEXP, KID_PREPARE_AGG, KID_PAIRS_AFF, KID_MILLER1, KID_F12_FOLD, KID_FINALEXP_ONE,
  KID_HASH, KID_ACCUM, KID_POINT_FOLD, KID_COMPRESS, KID_SIGN, KID_F12_IO, KID_MSM_SORT, KID_MSM_BUCKET, KID_MSM_CHUNK, KID_DECOMPRESS, KID_PAIRING_COOP, KID_COUNT
};
const char* KID_NAMES[KID_COUNT] = {"k_prepare", "k_miller2", "k_finalexp", "k_prepare_agg", "k_pairs_to_affine", "k_miller1", "k_f12_fold",
                                    "k_finalexp_one", "k_hash_to_point", "k_accumulate", "k_point_fold", "k_compress", "k_sign", "k_f12_io", "k_msm_sort", "k_msm_bucket", "k_msm_chunk", "k_decompress", "k_pairing_coop"};

struct Ctx {
  int dev = -1;
  hipStream_t stream = nullptr;
  uint8_t* arena = nullptr;
  size_t arena_cap = 0, arena_off = 0;
  std::mutex mu;
  hipEvent_t ev_host = nullptr;   // "the host may read what was copied so far" marker (verify_secure)
  hipStream_t side = nullptr;     // side stream: the message hash of a multi_verify tail runs beside the key sum
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // optional per-kernel timing with HIP events on `stream` (blsgpu_profile_*)
  bool prof_on = false;
  struct Pending { int kid; hipEvent_t e0, e1; };
  std::vector<Pending> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[KID_COUNT] = {0};
  uint64_t prof_cnt[KID_COUNT] = {0};
};
Ctx* g_ctx = nullptr;
std::mutex g_init_mu;

int fail(int code, const std::string& msg) {
  t_err = msg;
  return code;
}
#define HIPCK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      (void)hipGetLastError();                                                                     \
      return fail(BLSGPU_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                 \
    }                                                                                              \
  } while (0)

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ---- bump arena in device memory, reset per call
int arena_reserve(Ctx* c, size_t bytes) {
  if (bytes <= c->arena_cap) return 0;
  HIPCK(hipStreamSynchronize(c->stream));
  if (c->arena) HIPCK(hipFree(c->arena));
  c->arena = nullptr;
  c->arena_cap = 0;
  size_t cap = bytes + bytes / 8 + (1u << 20);
  HIPCK(hipMalloc((void**)&c->arena, cap));
  c->arena_cap = cap;
  return 0;
}
void* arena_take(Ctx* c, size_t bytes) {
  size_t off = (c->arena_off + 255) & ~(size_t)255;
  if (off + bytes > c->arena_cap) return nullptr;
  c->arena_off = off + bytes;
  return c->arena + off;
}
size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// device view of an input buffer: device pointers pass through, host buffers are staged into the arena
int stage_in(Ctx* c, const void* p, size_t bytes, const void** out) {
  if (bytes == 0) {
    *out = c->arena;
    return 0;
  }
  if (is_device_ptr(p)) {
    *out = p;
    return 0;
  }
  void* d = arena_take(c, bytes);
  if (!d) return fail(BLSGPU_E_HIP, "internal: arena too small");
  HIPCK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream));
  *out = d;
  return 0;
}
int copy_out(Ctx* c, void* dst, const void* dsrc, size_t bytes) {
  if (bytes == 0) return 0;
  HIPCK(hipMemcpyAsync(dst, dsrc, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  return 0;
}

size_t g1_size(int fmt) { return fmt == BLSGPU_FMT_RAW_PROJ ? 144 : fmt == BLSGPU_FMT_RAW_AFFINE ? 96 : 48; }
size_t g2_size(int fmt) { return fmt == BLSGPU_FMT_RAW_PROJ ? 288 : fmt == BLSGPU_FMT_RAW_AFFINE ? 192 : 96; }
size_t pk_size(int sg, int fmt) { return sg == 1 ? g2_size(fmt) : g1_size(fmt); }
size_t sig_size(int sg, int fmt) { return sg == 1 ? g1_size(fmt) : g2_size(fmt); }

const char* DST_TABLE[2][3] = {
    {"BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_", "BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_AUG_",
     "BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_"},  // reference src/impls/g1.rs:110,114,118
    {"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_", "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_AUG_",
     "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"},  // reference src/impls/g2.rs:108,112,116
};
dst_arg make_dst(const uint8_t* d, size_t n) {
  dst_arg a;
  memset(&a, 0, sizeof a);
  memcpy(a.b, d, n);
  a.len = (uint32_t)n;
  return a;
}
dst_arg scheme_dst(int sg, int scheme) {
  const char* s = DST_TABLE[sg - 1][scheme];
  return make_dst((const uint8_t*)s, strlen(s));
}

int check_common(int sg, int scheme, int fmt) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (sg != 1 && sg != 2) return fail(BLSGPU_E_ARG, "sig_group must be 1 (Bls12381G1Impl) or 2 (Bls12381G2Impl)");
  if (scheme < 0 || scheme > 2) return fail(BLSGPU_E_ARG, "scheme must be 0 (Basic), 1 (MessageAugmentation) or 2 (ProofOfPossession)");
  if (fmt != BLSGPU_FMT_RAW_PROJ && fmt != BLSGPU_FMT_RAW_AFFINE)
    return fail(BLSGPU_E_ARG, "this entry point takes BLSGPU_FMT_RAW_PROJ or BLSGPU_FMT_RAW_AFFINE points");
  return 0;
}

hipEvent_t prof_event(Ctx* c) {
  if (!c->prof_pool.empty()) {
    hipEvent_t e = c->prof_pool.back();
    c->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
void prof_pre(Ctx* c, int kid) {
  if (!c->prof_on) return;
  Ctx::Pending p{kid, prof_event(c), prof_event(c)};
  (void)hipEventRecord(p.e0, c->stream);
  c->prof_pending.push_back(p);
}
void prof_post(Ctx* c) {
  if (!c->prof_on) return;
  (void)hipEventRecord(c->prof_pending.back().e1, c->stream);
}
// call after the stream has been synchronised
void prof_flush(Ctx* c) {
  for (auto& p : c->prof_pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      c->prof_ms[p.kid] += ms;
      c->prof_cnt[p.kid]++;
    } else {
      (void)hipGetLastError();
    }
    c->prof_pool.push_back(p.e0);
    c->prof_pool.push_back(p.e1);
  }
  c->prof_pending.clear();
}
#define KL(kid, kern, grid, block, ...)                                         \
  do {                                                                          \
    prof_pre(c, kid);                                                           \
    hipLaunchKernelGGL(kern, grid, block, 0, c->stream, __VA_ARGS__);           \
    prof_post(c);                                                               \
  } while (0)
// one Miller loop per MILLER1_GROUP items: leaves ceil(cnt / MILLER1_GROUP) partial products at the start of the Fp12 workspace
#define MILLER1_OUTPUTS(cnt) (((cnt) + MILLER1_GROUP - 1) / MILLER1_GROUP)
#define MILLER1_LAUNCH(cnt, ...) KL(KID_MILLER1, k_miller1s, dim3(blocks_for(2 * MILLER1_OUTPUTS(cnt))), dim3(BLS_BLOCK), cnt, __VA_ARGS__)
#define SYNC_FLUSH(c)                          \
  do {                                         \
    HIPCK(hipStreamSynchronize((c)->stream));  \
    prof_flush(c);                             \
  } while (0)

// batches up to this size use the wave-cooperative pairing (one wave per item); BLSGPU_COOP_MAX overrides (0 = never)
size_t coop_max_items() {
  static long v = -1;
  if (v < 0) {
    const char* e = getenv("BLSGPU_COOP_MAX");
    v = e ? atol(e) : 6144;   // measured crossover with the lane-split kernels: 11.9 vs 13.7 ms at 6,144 items, 14.9 vs 13.8 ms at 8,192
    if (v < 0) v = 0;
  }
  return (size_t)v;
}

unsigned blocks_for(size_t n) { return (unsigned)((n + BLS_BLOCK - 1) / BLS_BLOCK); }

// ---- shared device pipelines (stream-ordered; caller holds the context mutex) ------------------------

// the two-pair pairing check of every item whose status is still BLS_OK: status <- OK / INVALID_SIGNATURE.
// fixed_g2: the second pair's G2 member is -g2 (precomputed lines)
int run_pairing2(Ctx* c, size_t n, uint32_t* d_pairs, uint32_t* d_f, int32_t* d_status, int fixed_g2) {
  if (n <= coop_max_items()) {  // small batches and single-item tails: one wave per item
    KL(KID_PAIRING_COOP, k_pairing_coop, dim3((unsigned)n), dim3(BLS_BLOCK), n, d_pairs, d_status, fixed_g2);
  } else {                      // two lanes per item (tower_split.cuh)
    KL(KID_MILLER2, k_miller2s, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), n, d_pairs, d_status, d_f, fixed_g2);
    KL(KID_FINALEXP, k_finalexps, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), n, d_f, d_status);
  }
  HIPCK(hipGetLastError());
  return 0;
}

// one core_verify per item: statuses end up in d_status (device)
int run_verify_items(Ctx* c, int sg, int aug, const uint8_t* d_pks, const uint8_t* d_sigs, int fmt, const uint8_t* d_msgs,
                     const uint64_t* d_offs, int single_msg, const dst_arg& dst, size_t n, uint32_t* d_pairs, uint32_t* d_f,
                     int32_t* d_status, int pre_status = 0) {
  if (n == 0) return 0;
  // two lanes per item (the two SSWU maps side by side, G2 point arithmetic on the lane-split tower): always for
  // Bls12381G2Impl, whose hash-to-G2 halves its per-lane work that way; for Bls12381G1Impl only in latency mode (the
  // Fp-only remainder would run redundantly and a full batch is faster with one lane per item -- both measured)
  const int two_lanes = (sg == 2 || n <= coop_max_items()) ? 1 : 0;
  unsigned nb = blocks_for(two_lanes ? 2 * n : n);
  if (sg == 1)
    KL(KID_PREPARE, k_prepare<1>, dim3(nb), dim3(BLS_BLOCK), n, d_pks, d_sigs, fmt, aug, d_msgs, d_offs, single_msg, dst, d_pairs, d_status, pre_status, two_lanes);
  else
    KL(KID_PREPARE, k_prepare<2>, dim3(nb), dim3(BLS_BLOCK), n, d_pks, d_sigs, fmt, aug, d_msgs, d_offs, single_msg, dst, d_pairs, d_status, pre_status, two_lanes);
  return run_pairing2(c, n, d_pairs, d_f, d_status, sg == 1 ? 1 : 0);
}

// product of m Fp12 values in a workspace (stride given) folded into item 0
int run_f12_fold(Ctx* c, uint32_t* d_f, size_t m, size_t stride) {
  size_t cur = m;
  while (cur > 1) {
    size_t half = (cur + 1) / 2;
    KL(KID_F12_FOLD, k_f12_fold, dim3(blocks_for(half)), dim3(BLS_BLOCK), cur, half, d_f, stride);
    cur = half;
  }
  HIPCK(hipGetLastError());
  return 0;
}
// ... then the verdict of item 0
int run_f12_product_verdict(Ctx* c, uint32_t* d_f, size_t m, size_t stride, int32_t* d_verdict) {
  int rc = run_f12_fold(c, d_f, m, stride);
  if (rc) return rc;
  if (coop_max_items() > 0) KL(KID_FINALEXP_ONE, k_finalexp_coop, dim3(1), dim3(BLS_BLOCK), d_f, stride, d_verdict);
  else KL(KID_FINALEXP_ONE, k_finalexp_ones, dim3(1), dim3(BLS_BLOCK), d_f, stride, d_verdict);
  HIPCK(hipGetLastError());
  return 0;
}

size_t accumulate_lanes(size_t n) {
  size_t t = n / 4;
  if (t < 64) t = 64;
  if (t > 65536) t = 65536;
  return (t + 63) & ~(size_t)63;
}

bool msm_use_pippenger(size_t n);
template <int G>
int run_msm_pippenger(Ctx* c, const uint8_t* d_pts, int fmt, const uint8_t* d_scalars, const uint32_t* d_perm, size_t n, uint8_t* d_out);

// sum (or sum of scalar multiples) of n points of group G into partials[0] (RAW_PROJ, device)
template <int G>
int run_point_sum(Ctx* c, const uint8_t* d_pts, int fmt, const uint8_t* d_scalars, const uint32_t* d_perm, size_t n,
                  uint8_t* d_partials, size_t T) {
  if (d_scalars && msm_use_pippenger(n)) return run_msm_pippenger<G>(c, d_pts, fmt, d_scalars, d_perm, n, d_partials);
  unsigned nb = blocks_for(T);
  if (d_scalars)
    KL(KID_ACCUM, (k_accumulate<G, 1>), dim3(nb), dim3(BLS_BLOCK), n, d_pts, fmt, d_scalars, d_perm, d_partials, T);
  else
    if (G == 2 && !d_perm) KL(KID_ACCUM, k_accumulate_g2s, dim3(blocks_for(2 * T)), dim3(BLS_BLOCK), n, d_pts, fmt, d_partials, T);
    else KL(KID_ACCUM, (k_accumulate<G, 0>), dim3(nb), dim3(BLS_BLOCK), n, d_pts, fmt, d_scalars, d_perm, d_partials, T);
  size_t cur = T;
  while (cur > 1) {
    size_t half = (cur + 1) / 2;
    if (G == 2) KL(KID_POINT_FOLD, k_point_fold_g2s, dim3(blocks_for(2 * half)), dim3(BLS_BLOCK), cur, half, d_partials);
    else KL(KID_POINT_FOLD, k_point_fold<G>, dim3(blocks_for(half)), dim3(BLS_BLOCK), cur, half, d_partials);
    cur = half;
  }
  HIPCK(hipGetLastError());
  return 0;
}

// Pippenger MSM into out[0] (RAW_PROJ, device, Z = 1).  Workspace comes from the arena (caller reserved msm_ws_bytes).
struct msm_plan {
  int c, W, clast, CH;
  size_t nb, nchunks;
};
msm_plan msm_make_plan(size_t n, int G) {
  msm_plan p;
  // measured at 65,536 points (tools/dbg/msm.py).  The chunk kernel is latency-bound (doubling chains) and the bucket
  // kernel throughput-bound, and a G2 addition costs about three G1 additions: G2 is best with 11-bit windows and 4-bucket
  // chunks (sort 0.4 + buckets 3.3 + chunks 2.9 ms against 0.6 + 2.7 + 3.9 with 12 / 8), G1 with 12 / 8 (0.6 + 1.4 + 2.5)
  p.c = n >= 16384 ? (G == 2 ? 11 : 12) : 8;
  p.CH = (n >= 16384 && G == 2) ? 4 : 8;   // short chunks: the 2^(c w) doublings of the chunk lanes dominate their critical path
  if (const char* e = getenv("BLSGPU_MSM_C")) p.c = atoi(e);        // tuning overrides (window bits, buckets per chunk lane)
  if (const char* e = getenv("BLSGPU_MSM_CH")) p.CH = atoi(e);
  p.W = 255 / p.c;                      // scalars are < r < 2^255
  p.clast = 255 - p.c * (p.W - 1);      // the last window takes the remainder: c <= clast < 2c
  p.nb = ((size_t)(p.W - 1) << p.c) + ((size_t)1 << p.clast);
  p.nchunks = p.nb / p.CH;
  return p;
}
size_t msm_ws_bytes(size_t n) {         // callers reserve for either group
  size_t need = 0;
  for (int G = 1; G <= 2; G++) {
    msm_plan p = msm_make_plan(n, G);
    size_t b = 3 * pad256(4 * p.nb) + pad256(4 * n * p.W) + pad256(288 * p.nb) + pad256(288 * p.nchunks) + 4096;
    if (b > need) need = b;
  }
  return need;
}
bool msm_use_pippenger(size_t n) {
  static int force_naive = -1;
  if (force_naive < 0) {
    const char* e = getenv("BLSGPU_MSM_NAIVE");
    force_naive = (e && e[0] == '1') ? 1 : 0;
  }
  return !force_naive && n >= 1024;
}
template <int G>
int run_msm_pippenger(Ctx* c, const uint8_t* d_pts, int fmt, const uint8_t* d_scalars, const uint32_t* d_perm, size_t n, uint8_t* d_out) {
  msm_plan p = msm_make_plan(n, G);
  uint32_t* d_cnt = (uint32_t*)arena_take(c, 4 * p.nb);
  uint32_t* d_off = (uint32_t*)arena_take(c, 4 * p.nb);
  uint32_t* d_cur = (uint32_t*)arena_take(c, 4 * p.nb);
  uint32_t* d_idx = (uint32_t*)arena_take(c, 4 * n * p.W);
  uint8_t* d_sums = (uint8_t*)arena_take(c, 288 * p.nb);
  uint8_t* d_part = (uint8_t*)arena_take(c, 288 * p.nchunks);
  if (!d_cnt || !d_off || !d_cur || !d_idx || !d_sums || !d_part) return fail(BLSGPU_E_HIP, "internal: arena too small");
  HIPCK(hipMemsetAsync(d_cnt, 0, 4 * p.nb, c->stream));
  HIPCK(hipMemsetAsync(d_cur, 0, 4 * p.nb, c->stream));
  KL(KID_MSM_SORT, k_msm_count, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_scalars, p.c, p.W, p.clast, d_cnt);
  KL(KID_MSM_SORT, k_msm_scan, dim3(1), dim3(BLS_BLOCK), p.nb, d_cnt, d_off);
  KL(KID_MSM_SORT, k_msm_fill, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_scalars, p.c, p.W, p.clast, d_off, d_cur, d_idx);
  if (G == 2) KL(KID_MSM_BUCKET, k_msm_bucket_g2s, dim3(blocks_for(2 * p.nb)), dim3(BLS_BLOCK), p.nb, d_pts, fmt, d_perm, d_cnt, d_off, d_idx, d_sums);
  else KL(KID_MSM_BUCKET, k_msm_bucket<G>, dim3(blocks_for(p.nb)), dim3(BLS_BLOCK), p.nb, d_pts, fmt, d_perm, d_cnt, d_off, d_idx, d_sums);
  if (G == 2) KL(KID_MSM_CHUNK, k_msm_chunk_g2q, dim3(blocks_for(4 * p.nchunks)), dim3(BLS_BLOCK), p.c, p.W, p.clast, p.CH, d_sums, d_part);
  else KL(KID_MSM_CHUNK, k_msm_chunk_g1p, dim3(blocks_for(2 * p.nchunks)), dim3(BLS_BLOCK), p.c, p.W, p.clast, p.CH, d_sums, d_part);
  size_t cur = p.nchunks;
  while (cur > 1) {
    size_t half = (cur + 1) / 2;
    if (G == 2) KL(KID_POINT_FOLD, k_point_fold_g2s, dim3(blocks_for(2 * half)), dim3(BLS_BLOCK), cur, half, d_part);
    else KL(KID_POINT_FOLD, k_point_fold<G>, dim3(blocks_for(half)), dim3(BLS_BLOCK), cur, half, d_part);
    cur = half;
  }
  KL(KID_MSM_CHUNK, k_normalize<G>, dim3(1), dim3(BLS_BLOCK), d_part);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(d_out, d_part, G == 1 ? 144 : 288, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// 256-bit big-endian hash -> scalar mod r, 32 bytes little-endian; returns false when the result is zero.
// Restates `int_BE(hash) mod r` of reference src/secure_aggregation.rs:61-100.
const uint32_t R_LIMBS[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
bool hash_to_scalar_le(const uint8_t hash[32], uint8_t out[32]) {
  uint32_t v[8];
  for (int i = 0; i < 8; i++) {
    const uint8_t* q = hash + 4 * (7 - i);
    v[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  }
  for (int round = 0; round < 3; round++) {  // hash < 2^256 < 3r
    uint32_t d[8];
    uint64_t bw = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t s = (uint64_t)v[i] - R_LIMBS[i] - bw;
      d[i] = (uint32_t)s;
      bw = (s >> 63) & 1;
    }
    if (bw) break;
    memcpy(v, d, sizeof v);
  }
  uint32_t nz = 0;
  for (int i = 0; i < 8; i++) {
    nz |= v[i];
    out[4 * i] = (uint8_t)v[i];
    out[4 * i + 1] = (uint8_t)(v[i] >> 8);
    out[4 * i + 2] = (uint8_t)(v[i] >> 16);
    out[4 * i + 3] = (uint8_t)(v[i] >> 24);
  }
  return nz != 0;
}

// host part of hash_public_keys_with_sorted (reference src/secure_aggregation.rs:41-103): stable sort by serialised
// bytes, H = SHA-256(concat), t_i = SHA-256(BE32(i) || H) mod r.  Returns BLSGPU_OK or BLSGPU_INVALID_COEFFICIENT.
int secure_coefficients_host(const uint8_t* kb, size_t n, size_t width, std::vector<uint32_t>& perm, std::vector<uint8_t>& scalars) {
  const bool trace = getenv("BLSGPU_HOST_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  perm.resize(n);
  std::iota(perm.begin(), perm.end(), 0u);
  // byte-lexicographic and stable (Rust's sort_by is stable): LSD radix sort (4 x 16 bits, stable) on the first 8 bytes
  // read as a big-endian word, then a stable comparison sort of every run of equal prefixes on the remaining bytes
  std::vector<uint64_t> pre(n);
  for (size_t i = 0; i < n; i++) {
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) v = (v << 8) | kb[i * width + k];
    pre[i] = v;
  }
  {
    std::vector<uint32_t> tmp(n), cnt(65536);
    for (int pass = 0; pass < 4; pass++) {
      const int sh = 16 * pass;
      std::fill(cnt.begin(), cnt.end(), 0u);
      for (size_t i = 0; i < n; i++) cnt[(pre[perm[i]] >> sh) & 0xffff]++;
      uint32_t run = 0;
      for (size_t d = 0; d < 65536; d++) {
        const uint32_t c0 = cnt[d];
        cnt[d] = run;
        run += c0;
      }
      for (size_t i = 0; i < n; i++) tmp[cnt[(pre[perm[i]] >> sh) & 0xffff]++] = perm[i];
      perm.swap(tmp);
    }
  }
  for (size_t a = 0; a < n;) {
    size_t b = a + 1;
    while (b < n && pre[perm[b]] == pre[perm[a]]) b++;
    if (b - a > 1)
      std::stable_sort(perm.begin() + a, perm.begin() + b, [&](uint32_t x, uint32_t y) {
        return memcmp(kb + (size_t)x * width + 8, kb + (size_t)y * width + 8, width - 8) < 0;
      });
    a = b;
  }
  const double t_sorted = now();
  host_sha256 h;
  for (size_t i = 0; i < n; i++) h.update(kb + (size_t)perm[i] * width, width);
  uint8_t H[32];
  h.final(H);
  const double t_hashed = now();
  scalars.resize(32 * n);
  // t_i = SHA256(BE32(i) || H): independent one-block hashes, spread over the host cores (the sorted-key hash above is one
  // sequential stream and stays on one core)
  unsigned nthr = std::thread::hardware_concurrency();
  if (nthr > 16) nthr = 16;
  if (nthr < 1 || n < 4096) nthr = 1;
  std::vector<int> bad(nthr, 0);
  auto work = [&](unsigned t) {
    const size_t lo = n * t / nthr, hi = n * (t + 1) / nthr;
    for (size_t i = lo; i < hi; i++) {
      uint8_t buf[36] = {(uint8_t)(i >> 24), (uint8_t)(i >> 16), (uint8_t)(i >> 8), (uint8_t)i};
      memcpy(buf + 4, H, 32);
      uint8_t d[32];
      host_sha256 hi2;
      hi2.update(buf, 36);
      hi2.final(d);
      if (!hash_to_scalar_le(d, &scalars[32 * i])) bad[t] = 1;
    }
  };
  if (nthr == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthr; t++) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
  for (unsigned t = 0; t < nthr; t++)
    if (bad[t]) return BLSGPU_INVALID_COEFFICIENT;
  if (trace) fprintf(stderr, "[blsgpu] secure coefficients n=%zu: sort %.2f ms, key hash %.2f ms, coefficient hashes %.2f ms\n", n,
                     t_sorted - t_start, t_hashed - t_sorted, now() - t_hashed);
  return BLSGPU_OK;
}

}  // namespace

// =========================================================================================================
extern "C" {

int blsgpu_init(int device) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (g_ctx) return 0;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return fail(BLSGPU_E_NO_DEVICE, "no HIP device available: libblsgpu has no CPU fallback");
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= ndev) return fail(BLSGPU_E_ARG, "device ordinal out of range");
  HIPCK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(BLSGPU_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
  Ctx* c = new Ctx();
  c->dev = device;
  HIPCK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  g_ctx = c;
  return 0;
}

void blsgpu_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (!g_ctx) return;
  (void)hipSetDevice(g_ctx->dev);
  (void)hipStreamSynchronize(g_ctx->stream);
  if (g_ctx->arena) (void)hipFree(g_ctx->arena);
  if (g_ctx->ev_host) (void)hipEventDestroy(g_ctx->ev_host);
  if (g_ctx->ev_fork) (void)hipEventDestroy(g_ctx->ev_fork);
  if (g_ctx->ev_join) (void)hipEventDestroy(g_ctx->ev_join);
  if (g_ctx->side) (void)hipStreamDestroy(g_ctx->side);
  (void)hipStreamDestroy(g_ctx->stream);
  delete g_ctx;
  g_ctx = nullptr;
}

int blsgpu_profile_enable(int on) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  g_ctx->prof_on = on != 0;
  for (int k = 0; k < KID_COUNT; k++) {
    g_ctx->prof_ms[k] = 0;
    g_ctx->prof_cnt[k] = 0;
  }
  return 0;
}
int blsgpu_profile_count(void) { return KID_COUNT; }
int blsgpu_profile_get(int kernel_id, char* name, size_t name_cap, double* total_ms, uint64_t* launches) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (kernel_id < 0 || kernel_id >= KID_COUNT) return fail(BLSGPU_E_ARG, "kernel id out of range");
  std::lock_guard<std::mutex> lk(g_ctx->mu);
  if (name && name_cap) {
    strncpy(name, KID_NAMES[kernel_id], name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (total_ms) *total_ms = g_ctx->prof_ms[kernel_id];
  if (launches) *launches = g_ctx->prof_cnt[kernel_id];
  return 0;
}

size_t blsgpu_last_error(char* buf, size_t cap) {
  if (buf && cap) {
    size_t k = std::min(cap - 1, t_err.size());
    memcpy(buf, t_err.data(), k);
    buf[k] = 0;
  }
  return t_err.size();
}

int blsgpu_verify_batch(int sig_group, int scheme, const void* pks, const void* sigs, const uint8_t* msgs,
                        const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status) {
  const bool wire = fmt == BLSGPU_FMT_COMPRESSED || fmt == BLSGPU_FMT_LEGACY;
  int rc = check_common(sig_group, scheme, wire ? BLSGPU_FMT_RAW_PROJ : fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!pks || !sigs || !msg_offsets || !status) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  // total message bytes: last offset (read it from wherever it lives)
  uint64_t total = 0;
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  else total = msg_offsets[n];
  const size_t pkb = pk_size(sig_group, fmt) * n, sgb = sig_size(sig_group, fmt) * n;
  size_t need = pad256(pkb) + pad256(sgb) + pad256(total) + pad256(8 * (n + 1)) + pad256(4 * n) + 2 * pad256((size_t)WS_PAIRS_WORDS * 4 * n) + 4096 +
                (wire ? pad256(288 * n) + pad256(144 * n) : 0);
  rc = arena_reserve(c, need);
  if (rc) return rc;
  c->arena_off = 0;
  const void *d_pks, *d_sigs, *d_msgs, *d_offs;
  if ((rc = stage_in(c, pks, pkb, &d_pks))) return rc;
  if ((rc = stage_in(c, sigs, sgb, &d_sigs))) return rc;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
  int32_t* d_status = (int32_t*)arena_take(c, 4 * n);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIRS_WORDS * 4 * n);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * n);
  if (!d_status || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
  int pre = 0, kfmt = fmt;
  if (wire) {
    // wire ingest: PublicKey / Signature from_bytes[_with_mode] -- checked decompression incl. subgroup
    // test, reference src/public_key.rs:58-74,158-171, src/signature.rs:231-253, src/impls/legacy.rs:100-170.
    // A decode failure is the item's status (key first, then signature), exactly what a caller that deserialises
    // before verifying would have seen.
    uint8_t* d_pk_raw = (uint8_t*)arena_take(c, (sig_group == 1 ? 288 : 144) * n);
    uint8_t* d_sig_raw = (uint8_t*)arena_take(c, (sig_group == 1 ? 144 : 288) * n);
    if (!d_pk_raw || !d_sig_raw) return fail(BLSGPU_E_HIP, "internal: arena too small");
    const int legacy = fmt == BLSGPU_FMT_LEGACY;
    if (sig_group == 1) {
      KL(KID_DECOMPRESS, k_decompress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, legacy, d_pk_raw, d_status, 0);
      KL(KID_DECOMPRESS, k_decompress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_sigs, legacy, d_sig_raw, d_status, 1);
    } else {
      KL(KID_DECOMPRESS, k_decompress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, legacy, d_pk_raw, d_status, 0);
      KL(KID_DECOMPRESS, k_decompress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_sigs, legacy, d_sig_raw, d_status, 1);
    }
    d_pks = d_pk_raw;
    d_sigs = d_sig_raw;
    pre = 1;
    kfmt = BLSGPU_FMT_RAW_PROJ;
  }
  rc = run_verify_items(c, sig_group, scheme == BLSGPU_SCHEME_AUG, (const uint8_t*)d_pks, (const uint8_t*)d_sigs, kfmt,
                        (const uint8_t*)d_msgs, (const uint64_t*)d_offs, 0, scheme_dst(sig_group, scheme), n, d_pairs, d_f, d_status, pre);
  if (rc) return rc;
  if ((rc = copy_out(c, status, d_status, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

// aug_prefix: MultiSignature::verify under MessageAugmentation prefixes the (aggregated) key bytes
// (reference src/traits/sig_aug.rs:20-24); verify_secure_message_augmentation does NOT, it only switches the DST
// (reference src/secure_aggregation.rs:236-246).
// d_hash != nullptr: H(msg) (RAW_PROJ, device) has already been computed on this stream (verify_secure overlaps it with the
// host's coefficient derivation); msg is then unused.
static int verify_one_tail(Ctx* c, int sig_group, int scheme, int aug_prefix, const uint8_t* d_pk_proj, const void* sig, int fmt,
                           const uint8_t* msg, size_t msg_len, int32_t* status, const uint8_t* d_hash = nullptr) {
  int rc;
  // bring the signature to RAW_PROJ next to the key so one k_prepare call (single fmt) serves both
  const void* d_sig_in;
  if ((rc = stage_in(c, sig, sig_size(sig_group, fmt), &d_sig_in))) return rc;
  const void* d_msg = nullptr;
  if (!d_hash && (rc = stage_in(c, msg, msg_len, &d_msg))) return rc;
  uint64_t offs_h[2] = {0, (uint64_t)msg_len};
  uint64_t* d_offs = (uint64_t*)arena_take(c, 16);
  int32_t* d_status = (int32_t*)arena_take(c, 4);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, WS_PAIRS_WORDS * 4);
  uint32_t* d_f = (uint32_t*)arena_take(c, WS_F_WORDS * 4);
  uint8_t* d_sig_proj = (uint8_t*)arena_take(c, 288);
  if (!d_offs || !d_status || !d_pairs || !d_f || !d_sig_proj) return fail(BLSGPU_E_HIP, "internal: arena too small");
  HIPCK(hipMemcpyAsync(d_offs, offs_h, 16, hipMemcpyHostToDevice, c->stream));
  // normalise the signature to RAW_PROJ with a 1-point "sum"
  if (sig_group == 1) rc = run_point_sum<1>(c, (const uint8_t*)d_sig_in, fmt, nullptr, nullptr, 1, d_sig_proj, 1);
  else rc = run_point_sum<2>(c, (const uint8_t*)d_sig_in, fmt, nullptr, nullptr, 1, d_sig_proj, 1);
  if (rc) return rc;
  if (d_hash) {
    if (sig_group == 1) KL(KID_PREPARE, k_prepare_hashed<1>, dim3(1), dim3(BLS_BLOCK), (size_t)1, d_pk_proj, (const uint8_t*)d_sig_proj, d_hash, d_pairs, d_status);
    else KL(KID_PREPARE, k_prepare_hashed<2>, dim3(1), dim3(BLS_BLOCK), (size_t)1, d_pk_proj, (const uint8_t*)d_sig_proj, d_hash, d_pairs, d_status);
    rc = run_pairing2(c, 1, d_pairs, d_f, d_status, sig_group == 1 ? 1 : 0);
  } else {
    rc = run_verify_items(c, sig_group, aug_prefix, d_pk_proj, d_sig_proj, BLSGPU_FMT_RAW_PROJ, (const uint8_t*)d_msg, d_offs, 1,
                          scheme_dst(sig_group, scheme), 1, d_pairs, d_f, d_status);
  }
  if (rc) return rc;
  if ((rc = copy_out(c, status, d_status, 4))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

int blsgpu_multi_verify(int sig_group, int scheme, const void* pks, size_t n, const void* sig, const uint8_t* msg,
                        size_t msg_len, int fmt, int32_t* status) {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!sig || !status || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t psz = pk_size(sig_group, fmt), T = accumulate_lanes(n);
  size_t need = pad256(psz * n) + pad256(288 * T) + 2 * pad256(msg_len) + 8192;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void* d_pks;
  if ((rc = stage_in(c, pks, psz * n, &d_pks))) return rc;
  uint8_t* d_part = (uint8_t*)arena_take(c, 288 * T);
  if (!d_part) return fail(BLSGPU_E_HIP, "internal: arena too small");
  // Unless the scheme prefixes the aggregated key to the message (MessageAugmentation, reference src/traits/sig_aug.rs:20-24),
  // H(msg) does not depend on the keys: hash it on a side stream (one wave) while the main stream sums the keys.
  uint8_t* d_hash = nullptr;
  if (scheme != BLSGPU_SCHEME_AUG && n > 0) {
    const void* d_msg0;
    if ((rc = stage_in(c, msg, msg_len, &d_msg0))) return rc;
    uint64_t* d_offs0 = (uint64_t*)arena_take(c, 16);
    d_hash = (uint8_t*)arena_take(c, 288);
    if (!d_offs0 || !d_hash) return fail(BLSGPU_E_HIP, "internal: arena too small");
    const uint64_t offs0[2] = {0, (uint64_t)msg_len};
    HIPCK(hipMemcpyAsync(d_offs0, offs0, 16, hipMemcpyHostToDevice, c->stream));
    if (!c->side) {
      HIPCK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
      HIPCK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      HIPCK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    HIPCK(hipEventRecord(c->ev_fork, c->stream));
    HIPCK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    if (sig_group == 1)
      hipLaunchKernelGGL(k_hash_to_g1, dim3(1), dim3(BLS_BLOCK), 0, c->side, (size_t)1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0,
                         scheme_dst(sig_group, scheme), d_hash, 1);
    else
      hipLaunchKernelGGL(k_hash_to_g2, dim3(1), dim3(BLS_BLOCK), 0, c->side, (size_t)1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0,
                         scheme_dst(sig_group, scheme), d_hash, 1);
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(c->ev_join, c->side));
  }
  // MultiPublicKey::from_public_keys: the serial `g += key` of reference src/traits/pk_multi.rs:7-13 as a tree sum
  if (sig_group == 1) rc = run_point_sum<2>(c, (const uint8_t*)d_pks, fmt, nullptr, nullptr, n, d_part, T);
  else rc = run_point_sum<1>(c, (const uint8_t*)d_pks, fmt, nullptr, nullptr, n, d_part, T);
  if (d_hash) {
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_join, 0);    // also on the error path: the side kernel reads the arena
    if (rc) (void)hipStreamSynchronize(c->side);
    if (e != hipSuccess) return fail(BLSGPU_E_HIP, "hipStreamWaitEvent failed");
  }
  if (rc) return rc;
  return verify_one_tail(c, sig_group, scheme, scheme == BLSGPU_SCHEME_AUG, d_part, sig, fmt, msg, msg_len, status, d_hash);
}

int blsgpu_aggregate_verify(int sig_group, int scheme, const void* pks, const uint8_t* msgs, const uint64_t* msg_offsets,
                            size_t n, const void* sig, int fmt, int32_t* status, uint64_t* aux) {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!sig || !status || !msg_offsets || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  uint64_t aux_h[2] = {0, 0};
  const bool trace = getenv("BLSGPU_HOST_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  // host views of offsets (and of messages for the Basic duplicate check)
  std::vector<uint64_t> offs_h(n + 1);
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(offs_h.data(), msg_offsets, 8 * (n + 1), hipMemcpyDeviceToHost));
  else memcpy(offs_h.data(), msg_offsets, 8 * (n + 1));
  const uint64_t total = offs_h[n];
  int32_t st = BLSGPU_OK;
  // reference src/traits/sig_basic.rs:46-58: first i whose message equals an earlier one.  In the reference this runs to
  // completion before anything else; here the GPU already hashes the messages to the curve meanwhile (the verdict keeps
  // the reference's precedence: a duplicate wins over whatever the device finds).
  std::vector<uint8_t> tmp;
  const uint8_t* mh = msgs;
  if (scheme == BLSGPU_SCHEME_BASIC && is_device_ptr(msgs)) {
    tmp.resize(total);
    HIPCK(hipMemcpy(tmp.data(), msgs, total, hipMemcpyDeviceToHost));
    mh = tmp.data();
  }
  auto duplicate_check = [&]() {
    // open-addressing table of message indices keyed by a 64-bit hash (computed on all host cores), equality verified on
    // the bytes; the scan is sequential so that the FIRST i with an earlier equal message is reported, as the reference does
    std::vector<uint64_t> hs(n);
    {
      unsigned nthr = std::thread::hardware_concurrency();
      if (nthr > 16) nthr = 16;
      if (nthr < 1 || n < 16384) nthr = 1;
      auto work = [&](unsigned t) {
        for (size_t i = n * t / nthr; i < n * (t + 1) / nthr; i++) {
          const uint8_t* p = mh + offs_h[i];
          size_t len = (size_t)(offs_h[i + 1] - offs_h[i]);
          uint64_t h = 0x9e3779b97f4a7c15ull ^ (len * 0xff51afd7ed558ccdull);
          while (len >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
            h ^= h >> 29;
            p += 8;
            len -= 8;
          }
          uint64_t w = 0;
          memcpy(&w, p, len);
          h = (h ^ w) * 0xff51afd7ed558ccdull;
          h ^= h >> 32;
          hs[i] = h;
        }
      };
      if (nthr == 1) {
        work(0);
      } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthr; t++) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
      }
    }
    size_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    std::vector<uint32_t> tab(cap, 0);          // index + 1, 0 = empty
    for (size_t i = 0; i < n && st == BLSGPU_OK; i++) {
      const size_t len = (size_t)(offs_h[i + 1] - offs_h[i]);
      size_t slot = (size_t)hs[i] & (cap - 1);
      while (tab[slot]) {
        const size_t j = tab[slot] - 1;
        if (hs[j] == hs[i] && (size_t)(offs_h[j + 1] - offs_h[j]) == len && memcmp(mh + offs_h[j], mh + offs_h[i], len) == 0) {
          st = BLSGPU_DUPLICATE_MESSAGE;
          aux_h[0] = j;
          aux_h[1] = i;
          break;
        }
        slot = (slot + 1) & (cap - 1);
      }
      if (st == BLSGPU_OK) tab[slot] = (uint32_t)(i + 1);
    }
  };
  const double t_dup = now();
  {
    const size_t m = n + 1, psz = pk_size(sig_group, fmt);
    size_t need = pad256(psz * n) + pad256(sig_size(sig_group, fmt)) + pad256(total) + pad256(8 * m) + pad256(4 * m) +
                  2 * pad256((size_t)WS_PAIRS_WORDS * 4 * m) + 8192;
    if ((rc = arena_reserve(c, need))) return rc;
    c->arena_off = 0;
    const void *d_pks, *d_sig, *d_msgs, *d_offs;
    if ((rc = stage_in(c, pks, psz * n, &d_pks))) return rc;
    if ((rc = stage_in(c, sig, sig_size(sig_group, fmt), &d_sig))) return rc;
    if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
    if ((rc = stage_in(c, msg_offsets, 8 * m, &d_offs))) return rc;
    const double t_staged = now();
    if (trace) fprintf(stderr, "[blsgpu] aggregate_verify n=%zu: host copies %.2f ms, staging %.2f ms\n", n, t_dup - t_start, t_staged - t_dup);
    int32_t* d_bad = (int32_t*)arena_take(c, 4 * m);
    int32_t* d_verdict = (int32_t*)arena_take(c, 4);
    uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIR1_WORDS * 4 * m);
    uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * m);
    if (!d_bad || !d_verdict || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
    dst_arg dst = scheme_dst(sig_group, scheme);
    int aug = scheme == BLSGPU_SCHEME_AUG;
    if (sig_group == 1)
      KL(KID_PREPARE_AGG, k_prepare_agg<1>, dim3(blocks_for(m)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_sig, fmt, aug,
                         (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_bad, 0);
    else
      KL(KID_PREPARE_AGG, k_prepare_agg<2>, dim3(blocks_for(2 * m)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_sig, fmt, aug,
                         (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_bad, 1);
    HIPCK(hipGetLastError());
    // The Miller loops and the final exponentiation are enqueued at once (flagged pairs are skipped on the device); the
    // identity flags come back together with the verdict, and the host applies the reference's precedence afterwards.
    MILLER1_LAUNCH(m, m, d_pairs, d_bad, d_f);
    if ((rc = run_f12_product_verdict(c, d_f, MILLER1_OUTPUTS(m), m, d_verdict))) return rc;
    if (scheme == BLSGPU_SCHEME_BASIC) {
      duplicate_check();          // on the host, while the device works
      if (trace) fprintf(stderr, "[blsgpu] aggregate_verify n=%zu: duplicate check %.2f ms (overlapped)\n", n, now() - t_staged);
    }
    std::vector<int32_t> bad(m);
    int32_t verdict = BLSGPU_OK;
    HIPCK(hipMemcpyAsync(bad.data(), d_bad, 4 * m, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipMemcpyAsync(&verdict, d_verdict, 4, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
    if (st != BLSGPU_OK) {
      // duplicate messages: reported before any identity check (sig_basic.rs:46-58 precedes core_aggregate_verify)
    } else if (bad[n]) {          // reference src/traits/sig_core.rs:155-167: signature identity first, ...
      st = BLSGPU_SIG_IDENTITY;
    } else {
      for (size_t i = 0; i < n; i++)
        if (bad[i]) {             // ... then the first identity key (1-based)
          st = BLSGPU_PK_IDENTITY;
          aux_h[0] = i + 1;
          break;
        }
      if (st == BLSGPU_OK) st = verdict;
    }
  }
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  if (aux) {
    if (is_device_ptr(aux)) HIPCK(hipMemcpy(aux, aux_h, 16, hipMemcpyHostToDevice));
    else memcpy(aux, aux_h, 16);
  }
  return 0;
}

int blsgpu_verify_secure(int sig_group, int scheme, const void* pks, size_t n, const void* sig, const uint8_t* msg,
                         size_t msg_len, int ser_format, int fmt, int32_t* status) {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!sig || !status || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  if (ser_format != 0 && ser_format != 1) return fail(BLSGPU_E_ARG, "ser_format must be 0 (Modern) or 1 (Legacy)");
  if (ser_format == 1 && sig_group != 2)
    return fail(BLSGPU_E_ARG, "Legacy serialization exists only for Bls12381G2Impl (48-byte keys), reference src/signature.rs:201-204");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t psz = pk_size(sig_group, fmt), width = sig_group == 1 ? 96 : 48, T = accumulate_lanes(n);
  size_t need = pad256(psz * n) + pad256(width * n) + pad256(4 * n) + pad256(32 * n) + pad256(288 * T) + pad256(msg_len) + 16384 + msm_ws_bytes(n);
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  int32_t st = BLSGPU_OK;
  if (n == 0) {
    // reference src/secure_aggregation.rs:189-195: Ok iff the signature is the identity
    const void* d_sig;
    if ((rc = stage_in(c, sig, sig_size(sig_group, fmt), &d_sig))) return rc;
    uint8_t* d_proj = (uint8_t*)arena_take(c, 288);
    std::vector<uint8_t> h(288);
    if (sig_group == 1) rc = run_point_sum<1>(c, (const uint8_t*)d_sig, fmt, nullptr, nullptr, 1, d_proj, 1);
    else rc = run_point_sum<2>(c, (const uint8_t*)d_sig, fmt, nullptr, nullptr, 1, d_proj, 1);
    if (rc) return rc;
    HIPCK(hipMemcpyAsync(h.data(), d_proj, 288, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
    const size_t zoff = sig_group == 1 ? 96 : 192, zlen = sig_group == 1 ? 48 : 96;
    bool inf = true;
    for (size_t k = 0; k < zlen; k++) inf = inf && h[zoff + k] == 0;
    st = inf ? BLSGPU_OK : BLSGPU_INVALID_SIGNATURE;
  } else {
    const void* d_pks;
    if ((rc = stage_in(c, pks, psz * n, &d_pks))) return rc;
    uint8_t* d_bytes = (uint8_t*)arena_take(c, width * n);
    uint32_t* d_perm = (uint32_t*)arena_take(c, 4 * n);
    uint8_t* d_scal = (uint8_t*)arena_take(c, 32 * n);
    uint8_t* d_part = (uint8_t*)arena_take(c, 288 * T);
    if (!d_bytes || !d_perm || !d_scal || !d_part) return fail(BLSGPU_E_HIP, "internal: arena too small");
    // PublicKey::to_bytes / to_bytes_with_mode of every key (reference src/secure_aggregation.rs:42,47; public_key.rs:146-151)
    if (sig_group == 1)
      KL(KID_COMPRESS, k_compress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
    else
      KL(KID_COMPRESS, k_compress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
    HIPCK(hipGetLastError());
    std::vector<uint8_t> kb(width * n);
    HIPCK(hipMemcpyAsync(kb.data(), d_bytes, width * n, hipMemcpyDeviceToHost, c->stream));
    // The host waits for the key bytes only; behind them the stream hashes the message to the curve (H(msg) does not depend
    // on the keys: verify_secure never prefixes them, reference src/secure_aggregation.rs:236-246), so the hash-to-curve of
    // the final core_verify runs while the host sorts and derives the coefficients.
    if (!c->ev_host) HIPCK(hipEventCreateWithFlags(&c->ev_host, hipEventDisableTiming));
    HIPCK(hipEventRecord(c->ev_host, c->stream));
    const void* d_msg0;
    if ((rc = stage_in(c, msg, msg_len, &d_msg0))) return rc;
    uint64_t* d_offs0 = (uint64_t*)arena_take(c, 16);
    uint8_t* d_hash = (uint8_t*)arena_take(c, 288);
    if (!d_offs0 || !d_hash) return fail(BLSGPU_E_HIP, "internal: arena too small");
    const uint64_t offs0[2] = {0, (uint64_t)msg_len};
    HIPCK(hipMemcpyAsync(d_offs0, offs0, 16, hipMemcpyHostToDevice, c->stream));
    if (sig_group == 1) KL(KID_HASH, k_hash_to_g1, dim3(1), dim3(BLS_BLOCK), (size_t)1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, scheme_dst(sig_group, scheme), d_hash, 1);
    else KL(KID_HASH, k_hash_to_g2, dim3(1), dim3(BLS_BLOCK), (size_t)1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, scheme_dst(sig_group, scheme), d_hash, 1);
    HIPCK(hipGetLastError());
    HIPCK(hipEventSynchronize(c->ev_host));
    std::vector<uint32_t> perm;
    std::vector<uint8_t> scal;
    st = secure_coefficients_host(kb.data(), n, width, perm, scal);
    if (st == BLSGPU_OK) {
      HIPCK(hipMemcpyAsync(d_perm, perm.data(), 4 * n, hipMemcpyHostToDevice, c->stream));
      HIPCK(hipMemcpyAsync(d_scal, scal.data(), 32 * n, hipMemcpyHostToDevice, c->stream));
      // aggregated_pk = sum t_i * pk_sorted[i]   (reference src/secure_aggregation.rs:201-204)
      if (sig_group == 1) rc = run_point_sum<2>(c, (const uint8_t*)d_pks, fmt, d_scal, d_perm, n, d_part, T);
      else rc = run_point_sum<1>(c, (const uint8_t*)d_pks, fmt, d_scal, d_perm, n, d_part, T);
      if (rc) return rc;
      return verify_one_tail(c, sig_group, scheme, 0, d_part, sig, fmt, msg, msg_len, status, d_hash);
    }
    SYNC_FLUSH(c);   // coefficient error: let the hash kernel (it reads the arena) drain before returning
  }
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  return 0;
}

int blsgpu_secure_coefficients(const uint8_t* key_bytes, size_t n, size_t width, uint32_t* out_perm, uint8_t* out_scalars,
                               int32_t* status) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (width != 48 && width != 96) return fail(BLSGPU_E_ARG, "width must be 48 or 96");
  if (!status || (n && (!key_bytes || !out_perm || !out_scalars))) return fail(BLSGPU_E_ARG, "null argument");
  std::vector<uint8_t> tmp;
  const uint8_t* kb = key_bytes;
  if (is_device_ptr(key_bytes)) {
    tmp.resize(n * width);
    HIPCK(hipMemcpy(tmp.data(), key_bytes, n * width, hipMemcpyDeviceToHost));
    kb = tmp.data();
  }
  std::vector<uint32_t> perm;
  std::vector<uint8_t> scal;
  int32_t st = secure_coefficients_host(kb, n, width, perm, scal);
  if (st == BLSGPU_OK && n) {
    HIPCK(hipMemcpy(out_perm, perm.data(), 4 * n, is_device_ptr(out_perm) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
    HIPCK(hipMemcpy(out_scalars, scal.data(), 32 * n, is_device_ptr(out_scalars) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  }
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  return 0;
}

static int hash_to_group(int group, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len, void* out) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (dst_len > 255) return fail(BLSGPU_E_ARG, "dst longer than 255 bytes is not supported");
  if (n == 0) return 0;
  if (!msg_offsets || !out || !dst) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  uint64_t total = 0;
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  else total = msg_offsets[n];
  const size_t osz = group == 1 ? 144 : 288;
  int rc = arena_reserve(c, pad256(total) + pad256(8 * (n + 1)) + pad256(osz * n) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void *d_msgs, *d_offs;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
  uint8_t* d_out = is_device_ptr(out) ? (uint8_t*)out : (uint8_t*)arena_take(c, osz * n);
  if (!d_out) return fail(BLSGPU_E_HIP, "internal: arena too small");
  dst_arg d = make_dst(dst, dst_len);
  // two lanes per message as in run_verify_items: always for G2, for G1 only in latency mode (small batches)
  const int two = (group == 2 || n <= coop_max_items()) ? 1 : 0;
  if (group == 1) KL(KID_HASH, k_hash_to_g1, dim3(blocks_for(two ? 2 * n : n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, d, d_out, two);
  else KL(KID_HASH, k_hash_to_g2, dim3(blocks_for(two ? 2 * n : n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, d, d_out, two);
  HIPCK(hipGetLastError());
  if (d_out != out && (rc = copy_out(c, out, d_out, osz * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
int blsgpu_hash_to_g1(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len, void* out) {
  return hash_to_group(1, msgs, msg_offsets, n, dst, dst_len, out);
}
int blsgpu_hash_to_g2(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len, void* out) {
  return hash_to_group(2, msgs, msg_offsets, n, dst, dst_len, out);
}

static int point_sum_entry(int group, const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (fmt != BLSGPU_FMT_RAW_PROJ && fmt != BLSGPU_FMT_RAW_AFFINE) return fail(BLSGPU_E_ARG, "fmt must be RAW_PROJ or RAW_AFFINE");
  if (!out || (n && !pts)) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t psz = group == 1 ? g1_size(fmt) : g2_size(fmt), osz = group == 1 ? 144 : 288, T = accumulate_lanes(n);
  int rc = arena_reserve(c, pad256(psz * n) + pad256(32 * n) + pad256(288 * T) + 4096 + (scalars ? msm_ws_bytes(n) : 0));
  if (rc) return rc;
  c->arena_off = 0;
  const void *d_pts, *d_scal = nullptr;
  if ((rc = stage_in(c, pts, psz * n, &d_pts))) return rc;
  if (scalars && (rc = stage_in(c, scalars, 32 * n, &d_scal))) return rc;
  uint8_t* d_part = (uint8_t*)arena_take(c, 288 * T);
  if (!d_part) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if (group == 1) rc = run_point_sum<1>(c, (const uint8_t*)d_pts, fmt, (const uint8_t*)d_scal, nullptr, n, d_part, T);
  else rc = run_point_sum<2>(c, (const uint8_t*)d_pts, fmt, (const uint8_t*)d_scal, nullptr, n, d_part, T);
  if (rc) return rc;
  if ((rc = copy_out(c, out, d_part, osz))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
int blsgpu_sum_g1(const void* pts, size_t n, int fmt, void* out) { return point_sum_entry(1, pts, nullptr, n, fmt, out); }
int blsgpu_sum_g2(const void* pts, size_t n, int fmt, void* out) { return point_sum_entry(2, pts, nullptr, n, fmt, out); }
int blsgpu_msm_g1(const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out) {
  if (n && !scalars) return fail(BLSGPU_E_ARG, "null scalars");
  return point_sum_entry(1, pts, scalars, n, fmt, out);
}
int blsgpu_msm_g2(const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out) {
  if (n && !scalars) return fail(BLSGPU_E_ARG, "null scalars");
  return point_sum_entry(2, pts, scalars, n, fmt, out);
}

int blsgpu_pairing_product_is_one(const void* g1s, const void* g2s, size_t n, int fmt, int32_t* is_one) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (fmt != BLSGPU_FMT_RAW_PROJ && fmt != BLSGPU_FMT_RAW_AFFINE) return fail(BLSGPU_E_ARG, "fmt must be RAW_PROJ or RAW_AFFINE");
  if (!is_one || (n && (!g1s || !g2s))) return fail(BLSGPU_E_ARG, "null argument");
  int32_t verdict = BLSGPU_OK;
  if (n > 0) {
    Ctx* c = g_ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCK(hipSetDevice(c->dev));
    int rc = arena_reserve(c, pad256(g1_size(fmt) * n) + pad256(g2_size(fmt) * n) + pad256(4 * n) + 2 * pad256((size_t)WS_F_WORDS * 4 * n) + 4096);
    if (rc) return rc;
    c->arena_off = 0;
    const void *d1, *d2;
    if ((rc = stage_in(c, g1s, g1_size(fmt) * n, &d1))) return rc;
    if ((rc = stage_in(c, g2s, g2_size(fmt) * n, &d2))) return rc;
    int32_t* d_skip = (int32_t*)arena_take(c, 4 * n + 4);
    uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIR1_WORDS * 4 * n);
    uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * n);
    if (!d_skip || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
    KL(KID_PAIRS_AFF, k_pairs_to_affine, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d1, (const uint8_t*)d2, fmt, d_pairs, d_skip);
    MILLER1_LAUNCH(n, n, d_pairs, d_skip, d_f);
    if ((rc = run_f12_product_verdict(c, d_f, MILLER1_OUTPUTS(n), n, d_skip + n))) return rc;
    HIPCK(hipMemcpyAsync(&verdict, d_skip + n, 4, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
  }
  int32_t one = verdict == BLSGPU_OK ? 1 : 0;
  if (is_device_ptr(is_one)) HIPCK(hipMemcpy(is_one, &one, 4, hipMemcpyHostToDevice));
  else *is_one = one;
  return 0;
}

int blsgpu_serialize(int group, const void* pts, size_t n, int fmt_in, int fmt_out, void* out, int32_t* status) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (group != 1 && group != 2) return fail(BLSGPU_E_ARG, "group must be 1 or 2");
  if ((fmt_in != BLSGPU_FMT_RAW_PROJ && fmt_in != BLSGPU_FMT_RAW_AFFINE) || (fmt_out != BLSGPU_FMT_COMPRESSED && fmt_out != BLSGPU_FMT_LEGACY))
    return fail(BLSGPU_E_ARG, "supported conversions: RAW_PROJ/RAW_AFFINE -> COMPRESSED/LEGACY");
  if (n == 0) return 0;
  if (!pts || !out) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t psz = group == 1 ? g1_size(fmt_in) : g2_size(fmt_in), osz = group == 1 ? 48 : 96;
  int rc = arena_reserve(c, pad256(psz * n) + pad256(osz * n) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void* d_pts;
  if ((rc = stage_in(c, pts, psz * n, &d_pts))) return rc;
  uint8_t* d_out = is_device_ptr(out) ? (uint8_t*)out : (uint8_t*)arena_take(c, osz * n);
  if (!d_out) return fail(BLSGPU_E_HIP, "internal: arena too small");
  int legacy = fmt_out == BLSGPU_FMT_LEGACY;
  if (group == 1) KL(KID_COMPRESS, k_compress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pts, fmt_in, legacy, d_out);
  else KL(KID_COMPRESS, k_compress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pts, fmt_in, legacy, d_out);
  HIPCK(hipGetLastError());
  if (d_out != out && (rc = copy_out(c, out, d_out, osz * n))) return rc;
  SYNC_FLUSH(c);
  if (status) {
    std::vector<int32_t> z(n, 0);
    HIPCK(hipMemcpy(status, z.data(), 4 * n, is_device_ptr(status) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  }
  return 0;
}

/* ProofOfPossession::verify for n (pk, proof) pairs: pop_verify(pk, sig) = core_verify(pk, sig, pk.to_bytes(), POP_DST)
 * (reference src/proof_of_possession.rs:79-81, src/traits/sig_pop.rs:67-70; POP_DST src/impls/g1.rs:119, g2.rs:117). */
int blsgpu_pop_verify_batch(int sig_group, const void* pks, const void* proofs, size_t n, int fmt, int32_t* status) {
  int rc = check_common(sig_group, 0, fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!pks || !proofs || !status) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t pkb = pk_size(sig_group, fmt) * n, sgb = sig_size(sig_group, fmt) * n;
  size_t need = pad256(pkb) + pad256(sgb) + pad256(4 * n) + 2 * pad256((size_t)WS_PAIRS_WORDS * 4 * n) + 4096;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d_pks, *d_sigs;
  if ((rc = stage_in(c, pks, pkb, &d_pks))) return rc;
  if ((rc = stage_in(c, proofs, sgb, &d_sigs))) return rc;
  uint64_t* d_offs = (uint64_t*)arena_take(c, 16);
  int32_t* d_status = (int32_t*)arena_take(c, 4 * n);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIRS_WORDS * 4 * n);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * n);
  if (!d_offs || !d_status || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
  HIPCK(hipMemsetAsync(d_offs, 0, 16, c->stream));
  const char* pd = sig_group == 1 ? "BLS_POP_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_" : "BLS_POP_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_";
  // mode 2: the message is the compressed key (single_msg = 1 keeps the unused message indexing in bounds)
  rc = run_verify_items(c, sig_group, 2, (const uint8_t*)d_pks, (const uint8_t*)d_sigs, fmt, (const uint8_t*)d_offs, d_offs, 1,
                        make_dst((const uint8_t*)pd, strlen(pd)), n, d_pairs, d_f, d_status);
  if (rc) return rc;
  if ((rc = copy_out(c, status, d_status, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

/* Sign-side secure aggregation: aggregate_secure[_with_mode] / AggregateSignature::from_signatures_secure
 * (reference src/secure_aggregation.rs:110-169,338-352, src/aggregate_signature.rs:191-227): sig_agg = sum t_i * sig[idx_i]
 * over the keys in sorted order, where idx_i is the FIRST input position whose serialised key equals the i-th sorted key
 * (the reference's `position` search, so duplicate keys pick the first matching signature).  n == 0: the identity. */
int blsgpu_aggregate_secure(int sig_group, const void* pks, const void* sigs, size_t n, int ser_format, int fmt, void* out_sig,
                            int32_t* status) {
  int rc = check_common(sig_group, 0, fmt);
  if (rc) return rc;
  if (!out_sig || !status || (n && (!pks || !sigs))) return fail(BLSGPU_E_ARG, "null argument");
  if (ser_format != 0 && ser_format != 1) return fail(BLSGPU_E_ARG, "ser_format must be 0 (Modern) or 1 (Legacy)");
  if (ser_format == 1 && sig_group != 2) return fail(BLSGPU_E_ARG, "Legacy serialization exists only for Bls12381G2Impl");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t psz = pk_size(sig_group, fmt), ssz = sig_size(sig_group, fmt), width = sig_group == 1 ? 96 : 48, T = accumulate_lanes(n);
  const size_t osz = sig_group == 1 ? 144 : 288;
  size_t need = pad256(psz * n) + pad256(ssz * n) + pad256(width * n) + pad256(4 * n) + pad256(32 * n) + pad256(288 * T) + 8192 + msm_ws_bytes(n);
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  int32_t st = BLSGPU_OK;
  uint8_t* d_part = (uint8_t*)arena_take(c, 288 * T);
  if (!d_part) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if (n == 0) {
    if (sig_group == 1) rc = run_point_sum<1>(c, nullptr, fmt, nullptr, nullptr, 0, d_part, T);
    else rc = run_point_sum<2>(c, nullptr, fmt, nullptr, nullptr, 0, d_part, T);
    if (rc) return rc;
  } else {
    const void *d_pks, *d_sigs;
    if ((rc = stage_in(c, pks, psz * n, &d_pks))) return rc;
    if ((rc = stage_in(c, sigs, ssz * n, &d_sigs))) return rc;
    uint8_t* d_bytes = (uint8_t*)arena_take(c, width * n);
    uint32_t* d_idx = (uint32_t*)arena_take(c, 4 * n);
    uint8_t* d_scal = (uint8_t*)arena_take(c, 32 * n);
    if (!d_bytes || !d_idx || !d_scal) return fail(BLSGPU_E_HIP, "internal: arena too small");
    if (sig_group == 1) KL(KID_COMPRESS, k_compress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
    else KL(KID_COMPRESS, k_compress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
    HIPCK(hipGetLastError());
    std::vector<uint8_t> kb(width * n);
    HIPCK(hipMemcpyAsync(kb.data(), d_bytes, width * n, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
    std::vector<uint32_t> perm;
    std::vector<uint8_t> scal;
    st = secure_coefficients_host(kb.data(), n, width, perm, scal);
    if (st == BLSGPU_OK) {
      std::vector<uint32_t> idx(n);
      for (size_t i = 0; i < n; i++)   // stable sort: equal keys keep input order, the first of a run has the smallest index
        idx[i] = (i > 0 && memcmp(&kb[(size_t)perm[i] * width], &kb[(size_t)perm[i - 1] * width], width) == 0) ? idx[i - 1] : perm[i];
      HIPCK(hipMemcpyAsync(d_idx, idx.data(), 4 * n, hipMemcpyHostToDevice, c->stream));
      HIPCK(hipMemcpyAsync(d_scal, scal.data(), 32 * n, hipMemcpyHostToDevice, c->stream));
      if (sig_group == 1) rc = run_point_sum<1>(c, (const uint8_t*)d_sigs, fmt, d_scal, d_idx, n, d_part, T);
      else rc = run_point_sum<2>(c, (const uint8_t*)d_sigs, fmt, d_scal, d_idx, n, d_part, T);
      if (rc) return rc;
      SYNC_FLUSH(c);                   // idx / scal are host vectors: finish the copies before they go out of scope
    }
  }
  if (st == BLSGPU_OK && (rc = copy_out(c, out_sig, d_part, osz))) return rc;
  SYNC_FLUSH(c);
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  return 0;
}

int blsgpu_deserialize(int group, const uint8_t* bytes, size_t n, int fmt_in, void* out, int32_t* status) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (group != 1 && group != 2) return fail(BLSGPU_E_ARG, "group must be 1 or 2");
  if (fmt_in != BLSGPU_FMT_COMPRESSED && fmt_in != BLSGPU_FMT_LEGACY) return fail(BLSGPU_E_ARG, "fmt_in must be COMPRESSED or LEGACY");
  if (n == 0) return 0;
  if (!bytes || !out || !status) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t isz = group == 1 ? 48 : 96, osz = group == 1 ? 144 : 288;
  int rc = arena_reserve(c, pad256(isz * n) + pad256(osz * n) + pad256(4 * n) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void* d_in;
  if ((rc = stage_in(c, bytes, isz * n, &d_in))) return rc;
  uint8_t* d_out = is_device_ptr(out) ? (uint8_t*)out : (uint8_t*)arena_take(c, osz * n);
  int32_t* d_st = is_device_ptr(status) ? status : (int32_t*)arena_take(c, 4 * n);
  if (!d_out || !d_st) return fail(BLSGPU_E_HIP, "internal: arena too small");
  const int legacy = fmt_in == BLSGPU_FMT_LEGACY;
  if (group == 1) KL(KID_DECOMPRESS, k_decompress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_in, legacy, d_out, d_st, 0);
  else KL(KID_DECOMPRESS, k_decompress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_in, legacy, d_out, d_st, 0);
  HIPCK(hipGetLastError());
  if (d_out != out && (rc = copy_out(c, out, d_out, osz * n))) return rc;
  if (d_st != status && (rc = copy_out(c, status, d_st, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

int blsgpu_sign_batch(int sig_group, int scheme, const uint8_t* sks, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n,
                      void* out_pks, void* out_sigs) {
  int rc = check_common(sig_group, scheme, BLSGPU_FMT_RAW_PROJ);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!sks || !msg_offsets || !out_pks || !out_sigs) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  uint64_t total = 0;
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  else total = msg_offsets[n];
  const size_t pkb = pk_size(sig_group, 0) * n, sgb = sig_size(sig_group, 0) * n;
  if ((rc = arena_reserve(c, pad256(32 * n) + pad256(total) + pad256(8 * (n + 1)) + pad256(pkb) + pad256(sgb) + 4096))) return rc;
  c->arena_off = 0;
  const void *d_sks, *d_msgs, *d_offs;
  if ((rc = stage_in(c, sks, 32 * n, &d_sks))) return rc;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
  uint8_t* d_pks = is_device_ptr(out_pks) ? (uint8_t*)out_pks : (uint8_t*)arena_take(c, pkb);
  uint8_t* d_sigs = is_device_ptr(out_sigs) ? (uint8_t*)out_sigs : (uint8_t*)arena_take(c, sgb);
  if (!d_pks || !d_sigs) return fail(BLSGPU_E_HIP, "internal: arena too small");
  dst_arg dst = scheme_dst(sig_group, scheme);
  int aug = scheme == BLSGPU_SCHEME_AUG;
  if (sig_group == 1)
    KL(KID_SIGN, k_sign<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_sks, aug, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pks, d_sigs);
  else
    KL(KID_SIGN, k_sign<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_sks, aug, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pks, d_sigs);
  HIPCK(hipGetLastError());
  if (d_pks != out_pks && (rc = copy_out(c, out_pks, d_pks, pkb))) return rc;
  if (d_sigs != out_sigs && (rc = copy_out(c, out_sigs, d_sigs, sgb))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

/* ---- sharded aggregate verify: the local part of core_aggregate_verify (reference
 * src/traits/sig_core.rs:149-178) for a contiguous shard of the (pk, msg) pairs.  out_f12 = product of the Miller
 * values of the shard's pairs [times the (sig, -g) pair when sig != NULL] BEFORE the final exponentiation, as a
 * 576-byte record; *first_bad = local index of the first identity public key, n when the signature is the identity,
 * or -1.  Duplicate-message detection is global and stays with the caller. */
int blsgpu_aggregate_partial(int sig_group, int scheme, const void* pks, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n,
                             const void* sig, int fmt, void* out_f12, int64_t* first_bad) {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!out_f12 || !first_bad || !msg_offsets || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  uint64_t total = 0;
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  else total = msg_offsets[n];
  const size_t m = n + 1, psz = pk_size(sig_group, fmt);
  size_t need = pad256(psz * n) + pad256(sig_size(sig_group, fmt)) + pad256(total) + pad256(8 * m) + pad256(4 * m) +
                2 * pad256((size_t)WS_PAIRS_WORDS * 4 * m) + 8192;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d_pks, *d_sig = nullptr, *d_msgs, *d_offs;
  if ((rc = stage_in(c, pks, psz * n, &d_pks))) return rc;
  if (sig && (rc = stage_in(c, sig, sig_size(sig_group, fmt), &d_sig))) return rc;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * m, &d_offs))) return rc;
  int32_t* d_bad = (int32_t*)arena_take(c, 4 * m);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIR1_WORDS * 4 * m);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * m);
  uint8_t* d_out = (uint8_t*)arena_take(c, 576);
  if (!d_bad || !d_pairs || !d_f || !d_out) return fail(BLSGPU_E_HIP, "internal: arena too small");
  dst_arg dst = scheme_dst(sig_group, scheme);
  int aug = scheme == BLSGPU_SCHEME_AUG;
  // without a signature the extra lane n is simply not launched (mm = n): k_prepare_agg treats i == n as the sig lane
  const size_t mm = sig ? m : n;
  std::vector<int32_t> bad(m, 0);
  int64_t fb = -1;
  if (mm > 0) {
    // the workspace stride is always n + 1 (k_prepare_agg's layout); lanes >= mm are never read
    if (sig_group == 1) {
      if (sig) KL(KID_PREPARE_AGG, k_prepare_agg<1>, dim3(blocks_for(m)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_sig, fmt, aug, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_bad, 0);
      else KL(KID_PREPARE_AGG, k_prepare_agg<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_pks, fmt, aug, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_bad, 0);
    } else {
      if (sig) KL(KID_PREPARE_AGG, k_prepare_agg<2>, dim3(blocks_for(2 * m)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_sig, fmt, aug, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_bad, 1);
      else KL(KID_PREPARE_AGG, k_prepare_agg<2>, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_pks, fmt, aug, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_bad, 1);
    }
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(bad.data(), d_bad, 4 * mm, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
    if (sig && bad[n]) fb = (int64_t)n;
    else
      for (size_t i = 0; i < n; i++)
        if (bad[i]) { fb = (int64_t)i; break; }
  }
  std::vector<uint8_t> one(576, 0);
  if (fb < 0 && mm > 0) {
    MILLER1_LAUNCH(mm, m, d_pairs, d_bad, d_f);   // stride n + 1 as k_prepare_agg wrote
  }
  *first_bad = fb;
  if (fb >= 0 || mm == 0) {
    // neutral element (Montgomery one in c0.a0) so that callers can always fold
    memcpy(one.data(), FP_ONE_HOST, 48);
    HIPCK(hipMemcpy(out_f12, one.data(), 576, is_device_ptr(out_f12) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
    return 0;
  }
  if ((rc = run_f12_fold(c, d_f, MILLER1_OUTPUTS(mm), m))) return rc;
  KL(KID_F12_IO, k_f12_export, dim3(1), dim3(BLS_BLOCK), d_f, m, d_out);
  HIPCK(hipGetLastError());
  if ((rc = copy_out(c, out_f12, d_out, 576))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

/* BlsSignatureCore::core_verify(pk, sig, msg, dst) with an explicit DST (reference src/traits/sig_core.rs:120-146):
 * n items share one DST; no message augmentation.  Used by the sharded verify_secure tail and by PoP checks
 * (pop_verify = core_verify(pk, sig, pk_bytes, POP_DST), src/traits/sig_pop.rs:67-70). */
static int core_verify_entry(int sig_group, const dst_arg& dst, int aug, const void* pks, const void* sigs, const uint8_t* msgs,
                             const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status) {
  int rc = 0;
  if (n == 0) return 0;
  if (!pks || !sigs || !msg_offsets || !status) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  uint64_t total = 0;
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  else total = msg_offsets[n];
  const size_t pkb = pk_size(sig_group, fmt) * n, sgb = sig_size(sig_group, fmt) * n;
  size_t need = pad256(pkb) + pad256(sgb) + pad256(total) + pad256(8 * (n + 1)) + pad256(4 * n) + 2 * pad256((size_t)WS_PAIRS_WORDS * 4 * n) + 4096;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d_pks, *d_sigs, *d_msgs, *d_offs;
  if ((rc = stage_in(c, pks, pkb, &d_pks))) return rc;
  if ((rc = stage_in(c, sigs, sgb, &d_sigs))) return rc;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
  int32_t* d_status = (int32_t*)arena_take(c, 4 * n);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIRS_WORDS * 4 * n);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * n);
  if (!d_status || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
  rc = run_verify_items(c, sig_group, aug, (const uint8_t*)d_pks, (const uint8_t*)d_sigs, fmt, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, 0,
                        dst, n, d_pairs, d_f, d_status);
  if (rc) return rc;
  if ((rc = copy_out(c, status, d_status, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
int blsgpu_core_verify(int sig_group, const uint8_t* dst, size_t dst_len, const void* pks, const void* sigs, const uint8_t* msgs,
                       const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status) {
  int rc = check_common(sig_group, 0, fmt);
  if (rc) return rc;
  if (dst_len > 255 || (!dst && dst_len)) return fail(BLSGPU_E_ARG, "dst must be at most 255 bytes");
  return core_verify_entry(sig_group, make_dst(dst, dst_len), 0, pks, sigs, msgs, msg_offsets, n, fmt, status);
}

/* SignCryptCiphertext::is_valid for n ciphertexts (reference src/sign_crypt_ciphertext.rs:86-101 -> BlsSignCrypt::valid,
 * src/traits/sign_crypt.rs:69-77): W' = H(U.to_bytes() || V) under the scheme's DST, then e(W, -g) * e(W', U) == 1 with
 * U and W not the identity.  That is core_verify(pk := U, sig := W, msg := U.to_bytes() || V, dst): the key-prefixed hash is
 * the augmentation path of k_prepare.  status[i] == 0 <=> valid (Choice 1); any other status <=> Choice 0. */
int blsgpu_signcrypt_valid_batch(int sig_group, int scheme, const void* us, const void* ws, const uint8_t* vs,
                                 const uint64_t* v_offsets, size_t n, int fmt, int32_t* status) {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  return core_verify_entry(sig_group, scheme_dst(sig_group, scheme), 1, us, ws, vs, v_offsets, n, fmt, status);
}

/* ProofOfKnowledge::verify for n proofs (reference src/proof_of_knowledge.rs:132-164 -> BlsSignatureProof::verify,
 * src/traits/sig_proof.rs:102-142; verify_timestamp_proof :145-175 derives y and calls the same check): status[i] in
 * {OK, COMMITMENT_IDENTITY, PROOF_IDENTITY, PK_IDENTITY, ZERO_CHALLENGE, INVALID_SIGNATURE (= BlsError::InvalidProof)}.
 * The message is hashed as given under the scheme's DST (the reference does not prefix the key here, even for
 * MessageAugmentation). */
int blsgpu_sig_proof_verify_batch(int sig_group, int scheme, const void* commitments, const void* proofs, const void* pks,
                                  const uint8_t* ys, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, int fmt,
                                  int32_t* status) {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!commitments || !proofs || !pks || !ys || !msg_offsets || !status) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  uint64_t total = 0;
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  else total = msg_offsets[n];
  const size_t pkb = pk_size(sig_group, fmt) * n, sgb = sig_size(sig_group, fmt) * n;
  size_t need = pad256(pkb) + 2 * pad256(sgb) + pad256(32 * n) + pad256(total) + pad256(8 * (n + 1)) + pad256(4 * n) +
                2 * pad256((size_t)WS_PAIRS_WORDS * 4 * n) + 4096;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d_u, *d_v, *d_pks, *d_ys, *d_msgs, *d_offs;
  if ((rc = stage_in(c, commitments, sgb, &d_u))) return rc;
  if ((rc = stage_in(c, proofs, sgb, &d_v))) return rc;
  if ((rc = stage_in(c, pks, pkb, &d_pks))) return rc;
  if ((rc = stage_in(c, ys, 32 * n, &d_ys))) return rc;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
  int32_t* d_status = (int32_t*)arena_take(c, 4 * n);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIRS_WORDS * 4 * n);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * n);
  if (!d_status || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
  const dst_arg dst = scheme_dst(sig_group, scheme);
  if (sig_group == 1)
    KL(KID_PREPARE, k_prepare_proof<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_u, (const uint8_t*)d_v, (const uint8_t*)d_pks,
       (const uint8_t*)d_ys, fmt, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_status);
  else
    KL(KID_PREPARE, k_prepare_proof<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_u, (const uint8_t*)d_v, (const uint8_t*)d_pks,
       (const uint8_t*)d_ys, fmt, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, dst, d_pairs, d_status);
  if ((rc = run_pairing2(c, n, d_pairs, d_f, d_status, sig_group == 1 ? 1 : 0))) return rc;
  if ((rc = copy_out(c, status, d_status, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

/* n independent two-pair checks  e(g1a[i], g2a[i]) * e(g1b[i], g2b[i]) == 1  -- Pairing::pairing(&[(..), (..)]).is_identity()
 * per item (reference src/traits/pairings.rs:50, glue src/helpers.rs:41-63), the shape of BlsSignCrypt::verify_share
 * (src/traits/sign_crypt.rs:192-207) and the ElGamal / time-lock share checks.  is_one[i] = 1 / 0.  Points must be in the
 * prime-order subgroups (every reference type guarantees it): a pair with an identity member contributes 1. */
int blsgpu_pairing2_check_batch(const void* g1a, const void* g2a, const void* g1b, const void* g2b, size_t n, int fmt,
                                int32_t* is_one) {
  int rc = check_common(1, 0, fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!g1a || !g2a || !g1b || !g2b || !is_one) return fail(BLSGPU_E_ARG, "null argument");
  Ctx* c = g_ctx;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCK(hipSetDevice(c->dev));
  const size_t b1 = (fmt == BLSGPU_FMT_RAW_PROJ ? 144 : 96) * n, b2 = (fmt == BLSGPU_FMT_RAW_PROJ ? 288 : 192) * n;
  size_t need = 2 * pad256(b1) + 2 * pad256(b2) + pad256(4 * n) + 2 * pad256((size_t)WS_PAIRS_WORDS * 4 * n) + 4096;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d1a, *d2a, *d1b, *d2b;
  if ((rc = stage_in(c, g1a, b1, &d1a))) return rc;
  if ((rc = stage_in(c, g2a, b2, &d2a))) return rc;
  if ((rc = stage_in(c, g1b, b1, &d1b))) return rc;
  if ((rc = stage_in(c, g2b, b2, &d2b))) return rc;
  int32_t* d_status = (int32_t*)arena_take(c, 4 * n);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIRS_WORDS * 4 * n);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * n);
  if (!d_status || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
  KL(KID_PAIRS_AFF, k_pairs2_to_affine, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d1a, (const uint8_t*)d2a, (const uint8_t*)d1b,
     (const uint8_t*)d2b, fmt, d_pairs, d_status);
  if ((rc = run_pairing2(c, n, d_pairs, d_f, d_status, 0))) return rc;
  KL(KID_PAIRS_AFF, k_status_to_flag, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_status);
  if ((rc = copy_out(c, is_one, d_status, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}

/* product of k Fp12 records (the partials of blsgpu_aggregate_partial) -> final exponentiation -> *is_one */
int blsgpu_fp12_product_is_one(const void* f12s, size_t k, int32_t* is_one) {
  if (!g_ctx) return fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)");
  if (!is_one || (k && !f12s)) return fail(BLSGPU_E_ARG, "null argument");
  int32_t verdict = BLSGPU_OK;
  if (k > 0) {
    Ctx* c = g_ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCK(hipSetDevice(c->dev));
    int rc = arena_reserve(c, pad256(576 * k) + pad256((size_t)WS_F_WORDS * 4 * k) + 4096);
    if (rc) return rc;
    c->arena_off = 0;
    const void* d_in;
    if ((rc = stage_in(c, f12s, 576 * k, &d_in))) return rc;
    uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * k);
    int32_t* d_v = (int32_t*)arena_take(c, 4);
    if (!d_f || !d_v) return fail(BLSGPU_E_HIP, "internal: arena too small");
    KL(KID_F12_IO, k_f12_import, dim3(blocks_for(k)), dim3(BLS_BLOCK), k, (const uint8_t*)d_in, d_f, k);
    if ((rc = run_f12_product_verdict(c, d_f, k, k, d_v))) return rc;
    HIPCK(hipMemcpyAsync(&verdict, d_v, 4, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
  }
  int32_t one = verdict == BLSGPU_OK ? 1 : 0;
  if (is_device_ptr(is_one)) HIPCK(hipMemcpy(is_one, &one, 4, hipMemcpyHostToDevice));
  else *is_one = one;
  return 0;
}

}  // extern "C"
