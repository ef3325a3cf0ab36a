// libblsgpu.so: host side of the C ABI declared in include/blsgpu.h.  HIP runtime only (no torch types).
// There is no CPU compute path in this library: without a gfx950 device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
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
thread_local size_t t_devidx = 0;    // which bound device the calling thread's leases come from (the sharded entry points set it)
thread_local bool t_nested = false;  // inside a sharded call: the per-device sub-calls must not shard again

enum {
  KID_PREPARE, KID_MILLER2, KID_FINALEXP, KID_PREPARE_AGG, KID_PAIRS_AFF, KID_MILLER1, KID_F12_FOLD, KID_FINALEXP_ONE,
  KID_HASH, KID_ACCUM, KID_POINT_FOLD, KID_COMPRESS, KID_SIGN, KID_F12_IO, KID_MSM_SORT, KID_MSM_BUCKET, KID_MSM_CHUNK, KID_DECOMPRESS, KID_PAIRING_COOP, KID_KEY_SORT, KID_COEFF, KID_DUP, KID_FIRST_BAD, KID_MSM_PREP, KID_MSM_NORM, KID_MSM_MERGE, KID_WIDE, KID_LINES, KID_CYCRUN,
  // round 4: one id per kernel on the paths bench.py reports (lumped ids made "the dominant kernel" meaningless)
  KID_MILLER2_V1, KID_FINALEXP_V1, KID_LINESP, KID_LINESP4, KID_LINE_QUAD, KID_F12_FOLD4, KID_F12_TREE, KID_HORNER, KID_MILLERFP, KID_PAIRING_POST, KID_PAIRING_PRE,
  KID_TAIL,        // everything launched on a context's tail stream: runs BESIDE the main stream's kernels, so its time is not additive
  KID_COUNT
};
const char* KID_NAMES[KID_COUNT] = {"k_prepare", "k_millerf2s", "k_finalexp2s", "k_prepare_agg", "k_pairs_to_affine", "k_miller1s", "k_f12_fold",
                                    "k_finalexp_one", "k_hash_to_point", "k_accumulate", "k_point_fold", "k_compress", "k_sign", "k_f12_io", "k_msm_sort", "k_msm_bucket", "k_msm_chunk", "k_decompress", "k_pairing_coop", "k_key_sort", "k_sha256_coeff", "k_duplicate_rule", "k_first_identity", "k_msm_prep", "k_normalize", "k_msm_merge", "k_wide", "k_lines2s", "k_cyc_run4",
                                    "k_miller2s", "k_finalexps", "k_linesp", "k_linesp4", "k_line_quad", "k_f12_fold4", "k_f12_tree_seg", "k_f12_horner_wide", "k_millerfp", "k_pairing_post", "k_pairing_pre",
                                    "tail_stream_overlapped"};

struct Ctx {
  int dev = -1;
  hipStream_t stream = nullptr;
  uint8_t* arena = nullptr;
  size_t arena_cap = 0, arena_off = 0;
  std::mutex mu;
  hipEvent_t ev_host = nullptr;   // "the host may read what was copied so far" marker (verify_secure)
  hipStream_t side = nullptr;     // side stream: the message hash of a multi_verify tail runs beside the key sum
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t side2 = nullptr;    // second side stream: the signature's half of a cut pairing check beside the message hash
  hipEvent_t ev_join2 = nullptr;
  uint8_t* hpin = nullptr;        // pinned host staging (sorted key bytes on their way to the host's SHA-256 stream)
  size_t hpin_cap = 0;
  uint8_t* hsmall = nullptr;      // pinned 64 KiB for the small host <-> device records of a call (offsets, flags, verdicts):
  size_t hsmall_off = 0;          // they outlive every early return, unlike stack variables
  hipStream_t tail = nullptr;     // the leftover items of a pairing product (run_miller_product) run here beside the main chunks
  hipEvent_t ev_tail_fork = nullptr, ev_tail_join = nullptr;
  uint32_t* fold_ws = nullptr;    // two small ping-pong buffers of the engine levels of an Fp12 fold tree (run_f12_fold)
  uint32_t* lines_ws = nullptr;   // merged line values on their way from k_lines2s to k_millerf2s (kernels.cuh): 19 KB per lane
  size_t lines_cap = 0;           // bytes; grown on demand, kept between calls
  uint32_t* stream_flags = nullptr;   // hand-over flags of the streamed cut (kernels.cuh k_pairing_stream): written by that kernel only
  uint32_t stream_epoch = 0;          // the value its last launch on this context published with
  // optional per-kernel timing with HIP events on `stream` (blsgpu_profile_*)
  bool prof_on = false;
  struct Pending { int kid; hipEvent_t e0, e1; };
  std::vector<Pending> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[KID_COUNT] = {0};
  uint64_t prof_cnt[KID_COUNT] = {0};
};
// One pool of contexts per bound device: a call leases a free context (own stream, own arena, own mutex), so concurrent
// callers overlap on the device instead of queueing behind one mutex.  Device index 0 is the device of blsgpu_init.
struct Device {
  int dev = -1;
  std::vector<Ctx*> pool;
  std::atomic<unsigned> rr{0};
};
std::vector<Device*> g_devices;     // written only under g_init_mu, before any compute call / after all of them
bool g_peer_ok = true;              // every bound device can read every other one's memory (needed to shard DEVICE buffers)
std::mutex g_init_mu;
bool initialised() { return !g_devices.empty(); }

void trim_workspaces(Ctx* c);
struct Lease {
  Ctx* c = nullptr;
  std::unique_lock<std::mutex> lk;
  Lease() = default;
  Lease(Lease&&) = default;
  // An entry point that fails half-way returns without having synchronised: drain what it enqueued before the caller's
  // buffers (and this context) are reused.  After a normal return both streams are idle and the queries cost nothing.
  ~Lease() {
    if (!c) return;
    if (hipStreamQuery(c->stream) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(c->stream);
    }
    if (c->side && hipStreamQuery(c->side) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(c->side);
    }
    if (c->side2 && hipStreamQuery(c->side2) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(c->side2);
    }
    if (c->tail && hipStreamQuery(c->tail) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(c->tail);
    }
    trim_workspaces(c);
  }
};
Lease acquire_ctx(size_t devidx) {
  Device* d = g_devices[devidx];
  Lease L;
  for (Ctx* c : d->pool) {
    std::unique_lock<std::mutex> lk(c->mu, std::try_to_lock);
    if (lk.owns_lock()) {
      L.c = c;
      L.lk = std::move(lk);
      return L;
    }
  }
  Ctx* c = d->pool[d->rr.fetch_add(1) % d->pool.size()];
  L.lk = std::unique_lock<std::mutex>(c->mu);
  L.c = c;
  return L;
}
#define CTX_ACQUIRE_ON(c, devidx)        \
  Lease lease_ = acquire_ctx(devidx);     \
  Ctx* c = lease_.c;                      \
  c->hsmall_off = 0;                      \
  HIPCK(hipSetDevice(c->dev))
#define CTX_ACQUIRE(c) CTX_ACQUIRE_ON(c, t_devidx)
#define NOT_INIT() fail(BLSGPU_E_NOT_INIT, "blsgpu_init has not been called (or found no gfx950 device)")
// no C++ exception may unwind into a C / Rust caller
#define API_CATCH                                                                              \
  catch (const std::bad_alloc&) { return fail(BLSGPU_E_HIP, "out of host memory"); }           \
  catch (const std::exception& e) { return fail(BLSGPU_E_HIP, std::string("internal: ") + e.what()); } \
  catch (...) { return fail(BLSGPU_E_HIP, "internal: unknown exception"); }

int fail(int code, const std::string& msg) {
  t_err = msg;
  return code;
}
// ---- run-time knobs: every BLSGPU_* environment variable the library reads, parsed ONCE into one struct (VERDICT r3 weak #9,
// next #7).  Two classes:
//   tuning  documented in README.md, clamped to what the kernels assume, change plans and sizes, never results;
//   A/B     select kernel forms kept for measurements and for the suite's agreement tests; honoured only together with
//           BLSGPU_AB_KNOBS=1, so that a single stray variable in somebody else's process cannot switch the product's kernels.
// BLSGPU_STRICT_ENV=1: blsgpu_init refuses to start when the environment holds a BLSGPU_ variable this table does not know, or an
// A/B variable without the gate (a typo or a leftover must not go unnoticed in production).
extern "C" char** environ;
struct Knobs {
  long coop_max = 4096, wide_max = 512, shard_min = 8192, contexts = 2, fake_devices = 0, acc_lanes = 57344;
  long msm_c = 0, msm_ch = 0, msm2_c = 0, msm2_ch = 0, msm2_q = 0;          // 0: the library's own choice
  long msm2_tables = 1;       // verify_secure: weighted window tables of every key, built while the host hashes (0: off)
  long stream_lines = 1;      // a check whose lines cannot be had early runs them BESIDE its Miller loop (k_pairing_stream; 0: two launches)
  long post_split = 3;        // the late part of a cut check runs its Miller loop on this many workgroups (k_pairing_post2: 3, 2; 0 or 1: one)
  long host_trace = 0, strict_env = 0, ab_knobs = 0;
  long ws_keep_mb = 4096;     // a context's line workspace above this many MiB is released when the call that grew it returns
  // A/B
  long miller_chunk = 65536, miller_v1 = 0, row_pad = 192, wide_mode = 2, finalexp_seg = 0, finalexp_v1 = 0, prepare_lanes = 0, product_tree = 1,
       tree_local = 0, lines4_max = -1, tree_engine_from = 0, agg_lanes = 0, msm_v1 = 0, msm_naive = 0, wide_test_block = 64;
  std::string problem;        // what BLSGPU_STRICT_ENV objects to (empty: nothing)
};
struct KnobSpec { const char* name; long Knobs::*field; long lo, hi; bool ab; };
const KnobSpec KNOB_TABLE[] = {
    {"BLSGPU_COOP_MAX", &Knobs::coop_max, 0, 1 << 20, false},          {"BLSGPU_WIDE_MAX", &Knobs::wide_max, 0, 4096, false},
    {"BLSGPU_SHARD_MIN", &Knobs::shard_min, 1, 1L << 40, false},        {"BLSGPU_CONTEXTS", &Knobs::contexts, 1, 16, false},
    {"BLSGPU_FAKE_DEVICES", &Knobs::fake_devices, 0, 64, false},        {"BLSGPU_ACC_LANES", &Knobs::acc_lanes, 64, 1 << 20, false},
    {"BLSGPU_MSM_C", &Knobs::msm_c, 4, 16, false},                      {"BLSGPU_MSM_CH", &Knobs::msm_ch, 1, 1 << 16, false},
    {"BLSGPU_MSM2_C", &Knobs::msm2_c, 4, 16, false},                    {"BLSGPU_MSM2_CH", &Knobs::msm2_ch, 1, 1 << 16, false},
    {"BLSGPU_MSM2_Q", &Knobs::msm2_q, 1, 8, false},                     {"BLSGPU_HOST_TRACE", &Knobs::host_trace, 0, 1, false},
    {"BLSGPU_WS_KEEP_MB", &Knobs::ws_keep_mb, 0, 1L << 20, false},        {"BLSGPU_MSM2_TABLES", &Knobs::msm2_tables, 0, 1, false},
    {"BLSGPU_STREAM_LINES", &Knobs::stream_lines, 0, 1, false},        {"BLSGPU_POST_SPLIT", &Knobs::post_split, 0, 3, false},
    {"BLSGPU_STRICT_ENV", &Knobs::strict_env, 0, 1, false},             {"BLSGPU_AB_KNOBS", &Knobs::ab_knobs, 0, 1, false},
    {"BLSGPU_MILLER_CHUNK", &Knobs::miller_chunk, 0, 65536, true},      {"BLSGPU_MILLER_V1", &Knobs::miller_v1, 0, 1, true},
    {"BLSGPU_ROW_PAD", &Knobs::row_pad, 0, 4096, true},                 {"BLSGPU_WIDE_MODE", &Knobs::wide_mode, 1, 2, true},
    {"BLSGPU_FINALEXP_SEG", &Knobs::finalexp_seg, 0, 1, true},          {"BLSGPU_FINALEXP_V1", &Knobs::finalexp_v1, 0, 1, true},
    {"BLSGPU_PREPARE_LANES", &Knobs::prepare_lanes, 0, 2, true},        {"BLSGPU_PRODUCT_TREE", &Knobs::product_tree, 0, 1, true},
    {"BLSGPU_TREE_LOCAL", &Knobs::tree_local, 0, 65536, true},          {"BLSGPU_LINES4_MAX", &Knobs::lines4_max, 0, 65536, true},
    {"BLSGPU_TREE_ENGINE_FROM", &Knobs::tree_engine_from, 0, 256, true}, {"BLSGPU_AGG_LANES", &Knobs::agg_lanes, 0, 2, true},
    {"BLSGPU_MSM_V1", &Knobs::msm_v1, 0, 1, true},                      {"BLSGPU_MSM_NAIVE", &Knobs::msm_naive, 0, 1, true},
    {"BLSGPU_WIDE_TEST_BLOCK", &Knobs::wide_test_block, 64, 256, true},
};
// read by the Python wrapper / debug builds only, never by this file's product paths: known names for the strict check
const char* const KNOB_OTHER_NAMES[] = {"BLSGPU_LIB", "BLSGPU_HASH_STOP"};
Knobs parse_knobs() {
  Knobs k;
  const char* gate = getenv("BLSGPU_AB_KNOBS");
  const bool ab_ok = gate && atol(gate) == 1;
  for (const KnobSpec& sp : KNOB_TABLE) {
    const char* e = getenv(sp.name);
    if (!e || !*e) continue;
    if (sp.ab && !ab_ok) {
      if (k.problem.empty()) k.problem = std::string(sp.name) + " is an A/B switch and needs BLSGPU_AB_KNOBS=1";
      fprintf(stderr, "libblsgpu: ignoring %s (an A/B switch: set BLSGPU_AB_KNOBS=1 to honour it)\n", sp.name);
      continue;
    }
    char* end = nullptr;
    long v = strtol(e, &end, 10);
    if (end == e || *end) {
      if (k.problem.empty()) k.problem = std::string(sp.name) + " is not an integer";
      continue;
    }
    k.*sp.field = v < sp.lo ? sp.lo : v > sp.hi ? sp.hi : v;     // clamped: a wild value cannot force a fallback by exhausting memory
  }
  for (char** e = environ; e && *e; e++) {
    if (strncmp(*e, "BLSGPU_", 7) != 0) continue;
    const char* eq = strchr(*e, '=');
    const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
    bool known = false;
    for (const KnobSpec& sp : KNOB_TABLE) known = known || name == sp.name;
    for (const char* o : KNOB_OTHER_NAMES) known = known || name == o;
    if (!known && k.problem.empty()) k.problem = "unknown variable " + name;
  }
  return k;
}
// Parsed when the library binds its devices (blsgpu_init / blsgpu_init_devices on an unbound library, under g_init_mu) and constant
// until blsgpu_shutdown: no compute call can run while the library is unbound, so nobody reads the struct while it is rewritten.
Knobs g_knobs;
const Knobs& knobs() { return g_knobs; }

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
size_t wide_max_items();
int arena_reserve(Ctx* c, size_t bytes) {
  bytes += (1u << 20) + wide_max_items() * (size_t)WREC_WORDS * 4;   // headroom that every call may rely on: the row-wide engine's per-item records
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

// small pinned records of a call (see Ctx::hsmall)
const size_t HSMALL_BYTES = 65536;
void* hsmall_take(Ctx* c, size_t bytes) {
  size_t off = (c->hsmall_off + 63) & ~(size_t)63;
  if (off + bytes > HSMALL_BYTES) return nullptr;
  c->hsmall_off = off + bytes;
  return c->hsmall + off;
}
// host -> device copy of a few bytes that live on the caller's stack
int h2d_small(Ctx* c, void* d, const void* src, size_t bytes) {
  void* h = hsmall_take(c, bytes);
  if (!h) return fail(BLSGPU_E_HIP, "internal: pinned record buffer exhausted");
  memcpy(h, src, bytes);
  HIPCK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}

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
// Several SMALL host buffers of one call (a single verification's key, signature, message, offsets) as ONE host-to-device copy
// through the context's pinned ring instead of one pageable copy each (~8 us apiece on the latency path).  Returns false when it
// does not apply (a device pointer among them, too large, ring full): the caller then stages one by one.
bool stage_in_packed(Ctx* c, const void* const* ptrs, const size_t* sizes, const void** outs, int k) {
  size_t total = 0;
  for (int j = 0; j < k; j++) {
    if (sizes[j] && is_device_ptr(ptrs[j])) return false;
    total += pad256(sizes[j]);
  }
  if (total == 0 || total > 16384) return false;
  uint8_t* h = (uint8_t*)hsmall_take(c, total);
  uint8_t* d = h ? (uint8_t*)arena_take(c, total) : nullptr;
  if (!h || !d) return false;
  size_t off = 0;
  for (int j = 0; j < k; j++) {
    if (sizes[j]) memcpy(h + off, ptrs[j], sizes[j]);
    outs[j] = d + off;
    off += pad256(sizes[j]);
  }
  if (hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}
int copy_out(Ctx* c, void* dst, const void* dsrc, size_t bytes) {
  if (bytes == 0) return 0;
  HIPCK(hipMemcpyAsync(dst, dsrc, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  return 0;
}

// a call's verdicts: copy, then wait for the stream.  A few verdicts for host memory travel through the pinned record buffer (a
// device-to-host copy into pageable memory makes the runtime stage and wait internally, ~10 us more on a 1.5 ms call).
void prof_flush(Ctx* c);
int copy_out_and_sync(Ctx* c, void* dst, const void* dsrc, size_t bytes) {
  void* h = nullptr;
  if (bytes && bytes <= 4096 && !is_device_ptr(dst) && (h = hsmall_take(c, bytes))) {
    HIPCK(hipMemcpyAsync(h, dsrc, bytes, hipMemcpyDeviceToHost, c->stream));
  } else {
    int rc = copy_out(c, dst, dsrc, bytes);
    if (rc) return rc;
  }
  HIPCK(hipStreamSynchronize(c->stream));
  prof_flush(c);
  if (h) memcpy(dst, h, bytes);
  return 0;
}

// the statuses of a call on their way to the caller.  A device-side hand-over that ran out of its bounded wait (kernels.cuh
// BLS_ERR_STREAM_TIMEOUT, the one negative value a kernel ever writes into a status) is a failure of the CALL, not a verdict:
// wherever the host gets to see the statuses it says so with the return code; a caller that keeps them on the device finds the
// -2 in the item's entry (include/blsgpu.h).
int status_out_and_sync(Ctx* c, int32_t* status, const int32_t* d_status, size_t n) {
  int rc = copy_out_and_sync(c, status, d_status, 4 * n);
  if (rc) return rc;
  if (n <= WPOST2_MAX_ITEMS && !is_device_ptr(status))
    for (size_t i = 0; i < n; i++)
      if (status[i] == BLS_ERR_STREAM_TIMEOUT) return fail(BLSGPU_E_HIP, "a workgroup hand-over on the device ran out of its bounded wait");
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
// RFC 9380 section 5.3.3: a domain separation tag longer than 255 bytes enters expand_message_xmd as
// SHA-256("H2C-OVERSIZE-DST-" || DST) -- the interface this replaces takes any DST (reference src/traits/hash_to_point.rs:11,
// src/impls/g1.rs:17-19, g2.rs:15-17: ExpandMsgXmd<Sha256> in the dependency applies the same rule)
dst_arg make_dst(const uint8_t* d, size_t n) {
  dst_arg a;
  memset(&a, 0, sizeof a);
  if (n > 255) {
    host_sha256 h;
    h.update((const uint8_t*)"H2C-OVERSIZE-DST-", 17);
    h.update(d, n);
    h.final(a.b);
    a.len = 32;
    return a;
  }
  if (n) memcpy(a.b, d, n);
  a.len = (uint32_t)n;
  return a;
}
dst_arg scheme_dst(int sg, int scheme) {
  const char* s = DST_TABLE[sg - 1][scheme];
  return make_dst((const uint8_t*)s, strlen(s));
}

int check_common(int sg, int scheme, int fmt) {
  if (!initialised()) return NOT_INIT();
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
void prof_pre(Ctx* c, int kid, hipStream_t s = nullptr) {
  if (!c->prof_on) return;
  Ctx::Pending p{kid, prof_event(c), prof_event(c)};
  (void)hipEventRecord(p.e0, s ? s : c->stream);
  c->prof_pending.push_back(p);
}
void prof_post(Ctx* c, hipStream_t s = nullptr) {
  if (!c->prof_on) return;
  (void)hipEventRecord(c->prof_pending.back().e1, s ? s : c->stream);
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
// a launch on another stream of the context (tail / side): recorded with events on THAT stream; the stream is joined to the main
// one before the call's final synchronisation, so prof_flush finds the events complete
#define KLS(kid, strm, kern, grid, block, ...)                                  \
  do {                                                                          \
    prof_pre(c, kid, strm);                                                     \
    hipLaunchKernelGGL(kern, grid, block, 0, strm, __VA_ARGS__);                \
    prof_post(c, strm);                                                         \
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
  // default 4096: measured crossover with the lane-split kernels (tools/dbg/mid3.py, round 3: whole calls from host lists): 11.4 vs 12.0 ms at
  // 3,072 items, 12.1 vs 12.1 at 4,096, 17.4 vs 12.5 at 6,144 (rounds 1-2, the one-kernel Miller loop: 6,144)
  return (size_t)knobs().coop_max;
}

// batches up to this size finish on the row-wide engine (one 256-thread workgroup per item; csrc/wide_engine.cuh):
// BLSGPU_WIDE_MAX overrides (0 = never).  One workgroup occupies the four SIMDs of a CU, so 256 items run side by side.
// measurement aid (tools/dbg/hash_phases.py): BLSGPU_HASH_STOP=k makes k_hash_to_g1_wide return after its k-th phase when it is
// launched through blsgpu_hash_to_g1 (the output is then meaningless); 0 / unset: the whole hash
int hash_phase_stop();
int hash_phase_stop() {
#if defined(BLS_DEBUG_KNOBS)        // measurement builds only (tools/dbg/hash_phases.py, hash2.py): a stray environment variable must
  const char* e = getenv("BLSGPU_HASH_STOP");   // not change hash outputs and verdicts of the product library
  return e ? atoi(e) & 0xff : 0;
#else
  return 0;
#endif
}
size_t wide_max_items() {
  // default 512: measured (tools/dbg/small.py): 1.9 ms up to 256 items (one workgroup per CU), 3.4 ms at 512, 5.3 ms at 768; the wave-cooperative path: 4.6-4.8 ms flat
  return (size_t)knobs().wide_max;
}

unsigned blocks_for(size_t n) { return (unsigned)((n + BLS_BLOCK - 1) / BLS_BLOCK); }

int pinned_reserve(Ctx* c, size_t bytes) {
  if (bytes <= c->hpin_cap) return 0;
  if (c->hpin) HIPCK(hipHostFree(c->hpin));
  c->hpin = nullptr;
  c->hpin_cap = 0;
  HIPCK(hipHostMalloc((void**)&c->hpin, bytes + bytes / 8 + 4096, hipHostMallocDefault));
  c->hpin_cap = bytes + bytes / 8 + 4096;
  return 0;
}

// exclusive prefix sum of m u32 counters (three launches, util_kernels.cuh); d_tiles: scan_tiles(m) words
size_t scan_tiles(size_t m) { return (m + UK_SCAN_TILE - 1) / UK_SCAN_TILE; }
int run_scan_u32(Ctx* c, int kid, size_t m, const uint32_t* d_in, uint32_t* d_out, uint32_t* d_tiles) {
  if (m == 0) return 0;
  const unsigned nt = (unsigned)scan_tiles(m);
  KL(kid, k_scan_tiles, dim3(nt), dim3(BLS_BLOCK), m, d_in, d_out, d_tiles);
  if (nt > 1) {
    KL(kid, k_scan_top, dim3(1), dim3(BLS_BLOCK), (size_t)nt, d_tiles);
    KL(kid, k_scan_apply, dim3(nt), dim3(BLS_BLOCK), m, d_out, (const uint32_t*)d_tiles);
  }
  HIPCK(hipGetLastError());
  return 0;
}
int run_scan_max_u32(Ctx* c, int kid, size_t m, uint32_t* d_v, uint32_t* d_tiles) {
  if (m == 0) return 0;
  const unsigned nt = (unsigned)scan_tiles(m);
  KL(kid, k_scan_max_tiles, dim3(nt), dim3(BLS_BLOCK), m, d_v, d_tiles);
  if (nt > 1) {
    KL(kid, k_scan_max_top, dim3(1), dim3(BLS_BLOCK), (size_t)nt, d_tiles);
    KL(kid, k_scan_max_apply, dim3(nt), dim3(BLS_BLOCK), m, d_v, (const uint32_t*)d_tiles);
  }
  HIPCK(hipGetLastError());
  return 0;
}



// ---- one process, several GPUs (blsgpu_init_devices): the items of a call are cut into contiguous ranges, one per bound
// device, each handled by a host thread whose leases come from that device's context pool
size_t shard_min_items() { return (size_t)knobs().shard_min; }   // BLSGPU_SHARD_MIN: below this many items a call stays on device 0
// how many devices a call over n items uses; device-resident inputs can only be sharded when the devices see each other
size_t shard_devices(size_t n, std::initializer_list<const void*> inputs) {
  if (t_nested || g_devices.size() < 2 || n < shard_min_items()) return 1;
  if (!g_peer_ok)
    for (const void* p : inputs)
      if (is_device_ptr(p)) return 1;
  size_t d = g_devices.size();
  while (d > 1 && n / d < shard_min_items() / 4) d--;
  return d;
}
// a helper thread that is joined when its owner goes out of scope, whichever way: an exception between the start of a helper and
// its join must unwind into API_CATCH, not into std::terminate of a process that holds the GPU (advisor finding, round 2)
struct JoinedThread {
  std::thread t;
  JoinedThread() = default;
  template <class F, class... A>
  explicit JoinedThread(F&& f, A&&... a) : t(std::forward<F>(f), std::forward<A>(a)...) {}
  JoinedThread(JoinedThread&&) = default;
  JoinedThread& operator=(JoinedThread&& o) {
    if (t.joinable()) t.join();
    t = std::move(o.t);
    return *this;
  }
  void join() { if (t.joinable()) t.join(); }
  ~JoinedThread() { if (t.joinable()) t.join(); }
};
// fn(d) for d < k on k host threads (thread d leases from device d); the first failure wins and its message is kept
template <class F>
int run_on_devices(size_t k, F&& fn) {
  std::vector<int> rcs(k, 0);
  std::vector<std::string> errs(k);
  std::vector<JoinedThread> th;        // joined on every way out, also when body(0) or an emplace_back throws
  auto body = [&](size_t d) {
    t_devidx = d;
    t_nested = true;
    try {
      rcs[d] = fn(d);
    } catch (const std::exception& e) {
      rcs[d] = fail(BLSGPU_E_HIP, std::string("internal: ") + e.what());
    } catch (...) {
      rcs[d] = fail(BLSGPU_E_HIP, "internal: unknown exception");
    }
    errs[d] = t_err;
  };
  th.reserve(k);
  for (size_t d = 1; d < k; d++) th.emplace_back(body, d);
  const size_t keep_dev = t_devidx;
  const bool keep_nested = t_nested;
  body(0);
  t_devidx = keep_dev;
  t_nested = keep_nested;
  for (auto& t : th) t.join();
  for (size_t d = 0; d < k; d++)
    if (rcs[d]) {
      t_err = errs[d];
      return rcs[d];
    }
  return 0;
}
struct NestedScope {     // the calling thread's own follow-up calls (fold, final verify) stay on device 0 and do not shard again
  bool keep;
  NestedScope() : keep(t_nested) { t_nested = true; }
  ~NestedScope() { t_nested = keep; }
};

// ---- shared device pipelines (stream-ordered; caller holds the context mutex) ------------------------

// items per pass of the two-kernel Miller loop (0: the one-kernel loop); one full machine round of lane pairs by default
size_t miller_chunk_items() { return knobs().miller_v1 ? 0 : (size_t)knobs().miller_chunk; }
size_t lanes_for(size_t items) { return (2 * items + BLS_BLOCK - 1) / BLS_BLOCK * BLS_BLOCK; }
// Row stride (in words) of the line workspace and the value store for that many lanes: NOT a power of two -- consecutive rows of
// a lane are a stride apart, and with 2^k-byte strides the 70 rows a wave touches per step camp on the same HBM channels
// (measured on the 262,144-lane chunks of config 4: BLSGPU_ROW_PAD=0 restores the bare stride for A/B runs)
size_t row_stride(size_t lanes) { return lanes + (size_t)knobs().row_pad; }
// the context's line workspace; 0 on success, non-zero (and no error recorded) when the device has no room for it
int lines_reserve(Ctx* c, size_t bytes) {
  if (bytes <= c->lines_cap) return 0;
  auto note = [&](const char* why) {      // the caller falls back to a slower plan: say so once per process instead of changing plans silently (advisor r3)
    static std::atomic<bool> said{false};
    if (!said.exchange(true)) fprintf(stderr, "libblsgpu: no room for a %zu MiB line workspace (%s): falling back to the one-kernel Miller loops\n", bytes >> 20, why);
  };
  if (hipStreamSynchronize(c->stream) != hipSuccess) {
    (void)hipGetLastError();
    note("stream error");
    return 1;
  }
  if (c->lines_ws) (void)hipFree(c->lines_ws);
  c->lines_ws = nullptr;
  c->lines_cap = 0;
  if (hipMalloc((void**)&c->lines_ws, bytes) != hipSuccess) {
    (void)hipGetLastError();
    c->lines_ws = nullptr;
    note("hipMalloc failed");
    return 1;
  }
  c->lines_cap = bytes;
  return 0;
}
// end of a call (Lease::~Lease, streams idle): a line workspace above BLSGPU_WS_KEEP_MB (default 4 GiB; the 65,536-item chunk of a
// batch needs 2.5 GB, the 262,144-pair plan of an aggregate 2.6 GB) does not stay with the pooled context for the life of the process
void trim_workspaces(Ctx* c) {
  if (c->lines_ws && (c->lines_cap >> 20) > (size_t)knobs().ws_keep_mb) {
    (void)hipFree(c->lines_ws);
    c->lines_ws = nullptr;
    c->lines_cap = 0;
  }
}

// the two-pair pairing check of every item whose status is still BLS_OK: status <- OK / INVALID_SIGNATURE.
// fixed_g2: the second pair's G2 member is a constant with precomputed lines: 1 = -g2, 2 = -[c] g2 (csrc/g2neg_lines.cuh)
int run_pairing2(Ctx* c, size_t n, uint32_t* d_pairs, uint32_t* d_f, int32_t* d_status, int fixed_g2) {
  if (n <= wide_max_items() && n <= coop_max_items()) {
    // single verifications and the one-verdict tails on the row-wide engine (csrc/wide_engine.cuh): one 256-thread workgroup
    // per item runs line coefficients, Miller loop, final exponentiation and verdict as one table program.
    // BLSGPU_WIDE_MODE=1: only the hard part of the final exponentiation on the engine, Miller loop + easy part per wave.
    if (knobs().wide_mode == 1) {
      uint32_t* d_easy = (uint32_t*)arena_take(c, (size_t)WIDE_EASY_WORDS * 4 * n);   // arena_reserve keeps 1 MiB of headroom: 768 B per item
      if (!d_easy) return fail(BLSGPU_E_HIP, "internal: arena too small");
      KL(KID_PAIRING_COOP, k_pairing_coop_easy, dim3((unsigned)n), dim3(BLS_BLOCK), n, (const uint32_t*)d_pairs, (const int32_t*)d_status, fixed_g2, d_easy);
      KL(KID_WIDE, k_finalexp_wide, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), n, (const uint32_t*)d_easy, d_status);
    } else {
      KL(KID_WIDE, k_pairing_wide, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), n, (const uint32_t*)d_pairs, d_status, fixed_g2);
    }
  } else if (n <= coop_max_items()) {  // small batches: one wave per item
    KL(KID_PAIRING_COOP, k_pairing_coop, dim3((unsigned)n), dim3(BLS_BLOCK), n, d_pairs, d_status, fixed_g2);
  } else {                      // two lanes per item (tower_split.cuh)
    // The Miller loop in two kernels (kernels.cuh k_lines2s / k_millerf2s), a chunk of at most miller_chunk_items() items at a
    // time: the chunk's merged line values pass through the context's line workspace.  BLSGPU_MILLER_V1=1, or no memory for
    // that workspace: the one-kernel loop of rounds 1 and 2.
    // BLSGPU_FINALEXP_SEG=1: the final exponentiation in six segments with the 63 compressed squarings of every a^x on FOUR lanes per
    // item at four waves per SIMD between them (k_cyc_run4).  Measured and not adopted (profiles/r03_pmc_finalexp_segments_cycrun4.json):
    // the squarings take 5.27 ms instead of ~5.6 ms inside k_finalexp2s, the six segments 4.28 ms: 9.55 ms against 9.49 ms for the one kernel.
    const bool finalexp_seg = knobs().finalexp_seg != 0;
    const bool finalexp_v1 = knobs().finalexp_v1 != 0;   // A/B: the one-kernel final exponentiation of rounds 1 and 2
    const size_t chunk = n < miller_chunk_items() ? n : miller_chunk_items();
    const size_t words_per_lane = (size_t)MILLER_ENTRIES * (LINE5_WORDS + (fixed_g2 ? 0 : LINE3_WORDS_H));   // two general pairs: pair 0's plain lines too
    if (chunk && lines_reserve(c, words_per_lane * 4 * row_stride(lanes_for(chunk))) == 0) {
      for (size_t first = 0; first < n; first += chunk) {
        const size_t cnt = n - first < chunk ? n - first : chunk;
        const size_t nlanes = lanes_for(cnt), lanes = row_stride(nlanes);       // `lanes` below: the row stride the kernels index with
        uint32_t* lines3 = c->lines_ws + (size_t)MILLER_ENTRIES * LINE5_WORDS * lanes;
        const dim3 grid((unsigned)(nlanes / BLS_BLOCK));
        if (fixed_g2) {
          KL(KID_LINES, k_lines2s, grid, dim3(BLS_BLOCK), n, first, cnt, (const uint32_t*)d_pairs, (const int32_t*)d_status, c->lines_ws, lines3, lanes, fixed_g2, 0);
        } else {
          KL(KID_LINES, k_lines2s, grid, dim3(BLS_BLOCK), n, first, cnt, (const uint32_t*)d_pairs, (const int32_t*)d_status, c->lines_ws, lines3, lanes, 0, 1);
          KL(KID_LINES, k_lines2s, grid, dim3(BLS_BLOCK), n, first, cnt, (const uint32_t*)d_pairs, (const int32_t*)d_status, c->lines_ws, lines3, lanes, 0, 2);
        }
        KL(KID_MILLER2, k_millerf2s, grid, dim3(BLS_BLOCK), n, first, cnt, (const int32_t*)d_status, (const uint32_t*)c->lines_ws, lanes, d_f);
        // the chunk's line values are consumed: the same memory is the value store of the final exponentiation (3.4 KB per lane)
        if (!finalexp_v1 && finalexp_seg) {
          // in segments, the compressed squarings of each a^x on four lanes per item at four waves per SIMD (k_cyc_run4)
          const dim3 grid4(blocks_for(4 * cnt));
          for (int seg = 0; seg <= 5; seg++) {
            KL(KID_FINALEXP, k_finalexp_seg, grid, dim3(BLS_BLOCK), seg, n, first, cnt, (const uint32_t*)d_f, c->lines_ws, lanes, d_status);
            if (seg < 5) KL(KID_CYCRUN, k_cyc_run4, grid4, dim3(BLS_BLOCK), cnt, c->lines_ws, lanes, (const int32_t*)d_status, first);
          }
        } else if (!finalexp_v1) {
          KL(KID_FINALEXP, k_finalexp2s, grid, dim3(BLS_BLOCK), n, first, cnt, (const uint32_t*)d_f, c->lines_ws, lanes, d_status);
        }
      }
      if (finalexp_v1) KL(KID_FINALEXP_V1, k_finalexps, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), n, d_f, d_status);
    } else {
      KL(KID_MILLER2_V1, k_miller2s, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), n, d_pairs, d_status, d_f, fixed_g2);
      KL(KID_FINALEXP_V1, k_finalexps, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), n, d_f, d_status);
    }
  }
  HIPCK(hipGetLastError());
  return 0;
}

// the context's side stream (created on first use) starts where the main stream is now
// which: bit 0 = mark the fork point on the main stream, bit 1 = make the side streams wait for it (a caller may enqueue the
// main stream's next kernel between the two: the host's enqueue time of the side work then is off that kernel's path)
int side_fork(Ctx* c, int which = 3) {
  if (!c->side) {
    HIPCK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIPCK(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming));
  }
  if (which & 1) HIPCK(hipEventRecord(c->ev_fork, c->stream));
  if (which & 2) {
    HIPCK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    HIPCK(hipStreamWaitEvent(c->side2, c->ev_fork, 0));
  }
  return 0;
}

// hash-to-G1 of a few messages on `stream`: one workgroup per message with the cofactor clearing on the engine while every
// message gets its own CU (up to 128 messages), one wave per message beyond that.  flags: bit 0 = every item hashes message 0.
void launch_hash_g1_small(hipStream_t stream, size_t n, const uint8_t* d_msgs, const uint64_t* d_offs, int flags, const dst_arg& dst, uint8_t* d_out,
                          uint32_t* d_rec) {
  if (n <= 128 && hash_phase_stop() == 0)     // measured (tools/dbg/small.py): at 256 items the pre-workgroups compete for the CUs and the one-wave hash wins
    hipLaunchKernelGGL(k_hash_to_g1_engine, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), 0, stream, n, d_msgs, d_offs, flags, dst, d_out, d_rec);
  else
    hipLaunchKernelGGL(k_hash_to_g1_wide, dim3((unsigned)n), dim3(BLS_BLOCK), 0, stream, n, d_msgs, d_offs, flags | hash_phase_stop() << 8, dst, d_out, d_rec);
}

// hash-to-G2 of a few messages on `stream` (RAW_PROJ out): one workgroup per message on the engine up to 128 messages; beyond
// that one WAVE per message for the maps and one workgroup per message for the engine program (d_pts: 768 bytes of scratch per
// message), or -- without scratch -- two lanes per message for the maps and one workgroup per point for the clearing.
bool launch_hash_g2_small(hipStream_t stream, size_t n, const uint8_t* d_msgs, const uint64_t* d_offs, int flags, const dst_arg& dst, uint8_t* d_out,
                          uint32_t* d_pts = nullptr, uint32_t* d_rec = nullptr) {
  // d_rec: the records of a cut check (WREC_Q0 of item i <- H(m_i)); true when they were written -- the lane-pair form only
  // fills d_out and the caller converts (k_prepare_keys part 4)
  if (n <= 128 && hash_phase_stop() == 0) {
    hipLaunchKernelGGL(k_hash_to_g2_engine, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), 0, stream, n, d_msgs, d_offs, flags, dst, d_out, d_rec);
    return d_rec != nullptr;
  }
  if (d_pts && hash_phase_stop() == 0) {
    hipLaunchKernelGGL(k_hash_to_g2_maps, dim3((unsigned)n), dim3(BLS_BLOCK), 0, stream, n, d_msgs, d_offs, flags, dst, d_pts);
    hipLaunchKernelGGL(k_g2_hash_tail_wide, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), 0, stream, n, (const uint32_t*)d_pts, d_out, d_rec);
    return d_rec != nullptr;
  }
  hipLaunchKernelGGL(k_hash_to_g2, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), 0, stream, n, d_msgs, d_offs, dst, d_out, 3);
  hipLaunchKernelGGL(k_g2_clear_wide, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), 0, stream, n, d_out);
  return false;
}

// The hand-over flags of the kernels whose workgroups pass data to each other inside one launch (kernels.cuh k_pairing_stream,
// k_pairing_post2): a buffer of the context that those kernels alone write, and a value per launch that no earlier launch used.
static int stream_epoch_next(Ctx* c) {
  const size_t fb = (size_t)WPOST2_MAX_ITEMS * WSTREAM_FLAGS * 4;
  if (!c->stream_flags) {
    HIPCK(hipMalloc((void**)&c->stream_flags, fb));
    c->stream_epoch = 0;
  }
  if (c->stream_epoch == 0 || c->stream_epoch == 0xffffffffu) {      // first use, or the counter is about to repeat itself
    HIPCK(hipMemsetAsync(c->stream_flags, 0, fb, c->stream));
    c->stream_epoch = 0;
  }
  c->stream_epoch++;
  return 0;
}
// the late part of the cut check alone (its lines are in the record): for up to WPOST2_MAX_ITEMS items with the Miller loop on two
// workgroups per item (kernels.cuh k_pairing_post2)
int launch_post(Ctx* c, size_t n, uint32_t* d_rec, int32_t* d_status) {
  if (knobs().post_split >= 2 && n <= WPOST2_MAX_ITEMS) {
    int rc = stream_epoch_next(c);
    if (rc) return rc;
    // three workgroups per item while every item can have three CUs at once, two up to WPOST2_MAX_ITEMS
    const unsigned parts = (knobs().post_split >= 3 && n <= WSTREAM_MAX_ITEMS) ? 3u : 2u;
    KL(KID_PAIRING_POST, k_pairing_post2, dim3((unsigned)n, parts), dim3(WIDE_ENGINE_BLOCK), n, d_rec, d_status, c->stream_flags, c->stream_epoch);
    return 0;
  }
  KL(KID_PAIRING_POST, k_pairing_post, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), n, (const uint32_t*)d_rec, d_status);
  return 0;
}
// Pair 0's lines and the late part of the cut check: ONE launch in which the lines travel from one workgroup to two others while
// the Miller loop already runs there (kernels.cuh k_pairing_stream) -- for up to WSTREAM_MAX_ITEMS checks whose lines have nothing
// to hide behind; otherwise the launches k_pairing_pre (part 0), then the late part as above.
int launch_lines_and_post(Ctx* c, size_t n, uint32_t* d_rec, int32_t* d_status) {
  if (knobs().stream_lines != 0 && n <= WSTREAM_MAX_ITEMS) {
    int rc = stream_epoch_next(c);
    if (rc) return rc;
    KL(KID_PAIRING_POST, k_pairing_stream, dim3((unsigned)n, 3), dim3(WIDE_ENGINE_BLOCK), n, d_rec, d_status, c->stream_flags, c->stream_epoch);
    return 0;
  }
  KL(KID_PAIRING_PRE, k_pairing_pre, dim3((unsigned)n, 1), dim3(WIDE_ENGINE_BLOCK), n, d_rec, (const int32_t*)d_status, 0, 0);
  return launch_post(c, n, d_rec, d_status);
}

// one core_verify per item: statuses end up in d_status (device)
int run_verify_items(Ctx* c, int sg, int aug, const uint8_t* d_pks, const uint8_t* d_sigs, int fmt, const uint8_t* d_msgs,
                     const uint64_t* d_offs, int single_msg, const dst_arg& dst, size_t n, uint32_t* d_pairs, uint32_t* d_f,
                     int32_t* d_status, int pre_status = 0) {
  if (n == 0) return 0;
  if (sg == 1 && aug == 0 && !pre_status && n <= wide_max_items() && n <= coop_max_items()) {
    // Single verifications of Bls12381G1Impl without a key prefix, cut where the inputs allow (csrc/kernels.cuh k_pairing_pre /
    // k_pairing_post): this stream hashes the messages (row-wide field type) while the side stream checks keys and
    // signatures, derives every key's line coefficients and runs the Miller loop of the (signature, -g2) pair -- two
    // workgroups per item, none of which needs H(m); what is left after the join is the Miller loop of (H(m), key) over
    // ready-made lines and the final exponentiation.
    uint32_t* d_rec = (uint32_t*)arena_take(c, (size_t)WREC_WORDS * 4 * n);
    if (!d_rec) return fail(BLSGPU_E_HIP, "internal: arena too small");
    // the LONG chain -- staging copy, hash, then the rest of the check -- stays on this stream and is enqueued first, so that
    // neither an event hand-over (~20 us each) nor the host's enqueue time of the side work sits on the critical path; the
    // side stream does what finishes early anyway
    int rc = side_fork(c, 1);
    if (rc) return rc;
    prof_pre(c, KID_HASH);
    launch_hash_g1_small(c->stream, n, d_msgs, d_offs, single_msg, dst, (uint8_t*)nullptr, d_rec);
    prof_post(c);
    if ((rc = side_fork(c, 2))) {
      (void)hipStreamSynchronize(c->stream);
      return rc;
    }
    hipLaunchKernelGGL(k_prepare_keys<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), 0, c->side, n, d_pks, d_sigs, (const uint8_t*)nullptr, fmt, 3, d_rec, d_status);
    hipLaunchKernelGGL(k_pairing_pre, dim3((unsigned)n, 2), dim3(WIDE_ENGINE_BLOCK), 0, c->side, n, d_rec, (const int32_t*)d_status, 0, 1);
    hipError_t e1 = hipGetLastError(), e2 = hipEventRecord(c->ev_join, c->side);
    hipError_t e3 = hipStreamWaitEvent(c->stream, c->ev_join, 0);      // also when something failed: the side kernels read the arena
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
      (void)hipStreamSynchronize(c->side);
      return fail(BLSGPU_E_HIP, "side-stream launch failed");
    }
    if ((rc = launch_post(c, n, d_rec, d_status))) {
      (void)hipStreamSynchronize(c->side);
      return rc;
    }
    HIPCK(hipGetLastError());
    return 0;
  }
  if (sg == 2 && aug == 0 && !pre_status && (!single_msg || n == 1) && n <= wide_max_items() && n <= coop_max_items()) {
    // The same cut for Bls12381G2Impl, pairs (key, H(m)) (-g1, signature): this stream carries the long chain -- the hash to G2
    // (row-wide maps, cofactor clearing on the engine), H(m)'s record and lines -- and is enqueued first; the side stream
    // checks keys and signatures and runs the signature's lines and Miller function; after the join the Miller function of
    // (key, H(m)) and the rest.
    uint32_t* d_rec = (uint32_t*)arena_take(c, (size_t)WREC_WORDS * 4 * n);
    uint8_t* d_hashes = (uint8_t*)arena_take(c, 288 * n);
    uint32_t* d_maps = n > 128 ? (uint32_t*)arena_take(c, 768 * n) : nullptr;      // nullptr (arena full): the lane-pair maps
    if (!d_rec || !d_hashes) return fail(BLSGPU_E_HIP, "internal: arena too small");
    int rc = side_fork(c, 1);
    if (rc) return rc;
    prof_pre(c, KID_HASH);
    const bool rec_done = launch_hash_g2_small(c->stream, n, d_msgs, d_offs, 0, dst, d_hashes, d_maps, d_rec);
    prof_post(c);
    if (!rec_done)
      KL(KID_PREPARE, k_prepare_keys<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)nullptr, (const uint8_t*)nullptr, (const uint8_t*)d_hashes, 0, 4,
         d_rec, d_status);
    if ((rc = side_fork(c, 2))) {
      (void)hipStreamSynchronize(c->stream);
      return rc;
    }
    hipLaunchKernelGGL(k_prepare_keys<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), 0, c->side, n, d_pks, d_sigs, (const uint8_t*)nullptr, fmt, 3, d_rec, d_status);
    // the lines of H(m) skip items that failed the identity checks: this stream waits for the statuses written just above
    hipError_t e1 = hipGetLastError(), e2 = hipEventRecord(c->ev_join2, c->side);
    hipLaunchKernelGGL(k_pairing_pre, dim3((unsigned)n, 1), dim3(WIDE_ENGINE_BLOCK), 0, c->side, n, d_rec, (const int32_t*)d_status, 2, 2);
    hipError_t e3 = hipGetLastError(), e4 = hipEventRecord(c->ev_join, c->side);
    hipError_t e5 = hipStreamWaitEvent(c->stream, c->ev_join2, 0);
    hipError_t e6 = hipStreamWaitEvent(c->stream, c->ev_join, 0);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess || e6 != hipSuccess) {
      (void)hipStreamSynchronize(c->side);
      return fail(BLSGPU_E_HIP, "side-stream launch failed");
    }
    // H(m)'s lines have nothing left to hide behind (the signature's half on the side stream ended under the hash): they run
    // beside the Miller loop that consumes them
    if ((rc = launch_lines_and_post(c, n, d_rec, d_status))) {
      (void)hipStreamSynchronize(c->side);
      return rc;
    }
    HIPCK(hipGetLastError());
    return 0;
  }
  if (sg == 1 && aug == 0 && !pre_status && n > wide_max_items() && n <= 1024 && n <= coop_max_items() && wide_max_items() > 0) {
    // 513 to 1,024 items (measured: 4.1 / 4.6 ms at 768 / 1,024 against 4.8 / 5.1; at 2,048 the two-lane hash wins): the wave-cooperative pairing, but with the one-wave row-wide hash (0.74 ms while every
    // message has a SIMD to itself, against 1.6 ms for the two-lane hash inside k_prepare) and the shared to-affine inversion after it
    uint8_t* d_hashes = (uint8_t*)arena_take(c, 144 * n);
    if (!d_hashes) return fail(BLSGPU_E_HIP, "internal: arena too small");
    // the hashes stay uncleared (bit 1) and the fixed pair is (sig, -[c] g2) (table 2), as everywhere for this implementation
    KL(KID_HASH, k_hash_to_g1_wide, dim3((unsigned)n), dim3(BLS_BLOCK), n, d_msgs, d_offs, (single_msg & 1) | 2, dst, d_hashes, (uint32_t*)nullptr);
    KL(KID_PREPARE, k_prepare_hashed<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_pks, d_sigs, (const uint8_t*)d_hashes, d_pairs, d_status, fmt);
    return run_pairing2(c, n, d_pairs, d_f, d_status, 2);
  }
  // two lanes per item (the two SSWU maps side by side, G2 point arithmetic on the lane-split tower) for both implementations
  // and every batch size (rounds 1-2 used one lane per item for full Bls12381G1Impl batches; see below)
  const int force_lanes = (int)knobs().prepare_lanes;   // A/B aid: 1 or 2 lanes per item in k_prepare<1>
  // two lanes per item also for full Bls12381G1Impl batches since round 3: 1.69 against 1.80 ms at 65,536 items (k_prepare is a one-wave-per-SIMD kernel
  // otherwise, and a lone wave issues a multiply-add every 8.8 cycles where two waves share the pipe at 4.4)
  const int two_lanes = force_lanes == 1 && sg == 1 ? 0 : 1;
  unsigned nb = blocks_for(two_lanes ? 2 * n : n);
  // Bls12381G1Impl: the message points stay UNCLEARED (a third of the hash) and the second pair is (sig, -[c] g2), c = h_eff^-1
  // mod r, whose line table is as constant as -g2's: e(h P', pk) e(sig, -g2) = 1  <=>  e(P', pk) e(sig, -[c] g2) = 1
  // (csrc/g2neg_lines.cuh; fixed_g2 = 2 selects that table)
  if (sg == 1)
    KL(KID_PREPARE, k_prepare<1>, dim3(nb), dim3(BLS_BLOCK), n, d_pks, d_sigs, fmt, aug, d_msgs, d_offs, single_msg, dst, d_pairs, d_status, pre_status, two_lanes | 2);
  else
    KL(KID_PREPARE, k_prepare<2>, dim3(nb), dim3(BLS_BLOCK), n, d_pks, d_sigs, fmt, aug, d_msgs, d_offs, single_msg, dst, d_pairs, d_status, pre_status, two_lanes);
  return run_pairing2(c, n, d_pairs, d_f, d_status, sg == 1 ? 2 : 0);
}

// Pairing products, second form (round 3): the product over the items entry by entry, then one Horner chain (kernels.cuh
// k_line_quad).  Chunks of one machine round of lane pairs walk their G2 points (k_linesp pass 1, or k_linesp4 on four lanes per
// item: the plain line values of every item and entry), k_line_quad multiplies four items' values per entry into an Fp12 value,
// k_f12_fold4 folds each entry's values four to one -- chunk by chunk while a level still fills the machine, then over all chunks'
// values together down to 128 per entry -- the engine takes those in two launches (k_f12_tree_seg) and runs the Horner chain over
// the 68 products (k_f12_horner_wide).  What is
// left beyond whole rounds (< 1,024 items) is a small chunk of its own on the tail stream, beside the others.  Leaves the
// Miller product as item 0 of d_f (*outputs = 1).  *done = false: not applicable (too few items,
// no engine, no memory, BLSGPU_PRODUCT_TREE=0) -- the caller falls back to the accumulator kernels.
// last_mode: what item mm - 1 is.  0: an item like the others.  1 / 2: (P, -g2) / (P, -[c] g2), the signature's pair of an
// aggregate verification of Bls12381G1Impl: its line values come from the fixed argument's table (k_lines_fixed), no point walk.
// 3: the signature's pair of Bls12381G2Impl, a general pair.  The pairs proper of an aggregate are counts like 32,768 or 65,536, and
// the line kernels have cliffs exactly there (the 1,025th wave puts two waves on one SIMD, which then take 1.7 times as long; the
// 2,049th runs alone after the others): in modes 1 - 3 the signature's pair is a one-item chunk of its own on the tail stream.
int run_miller_product_tree(Ctx* c, size_t mm, size_t stride, uint32_t* d_pairs, int32_t* d_bad, uint32_t* d_f, size_t* outputs, bool* done, int last_mode) {
  *done = false;
  const int enabled = (int)knobs().product_tree;
  const long local_env = knobs().tree_local;
  const long lines4_env = knobs().lines4_max;   // A/B: 0 = the two-lane line kernel everywhere
  if (!enabled || mm < 64 || !miller_chunk_items() || !(wide_max_items() > 0 && coop_max_items() > 0)) return 0;
  const size_t round = 65536, E = MILLER_ENTRIES;
  const size_t LOCAL = local_env > 16 ? (size_t)local_env : 1024;    // 68 x 1,024 lane pairs: one machine round
  // The line kernel of a chunk.  Four lanes per item (k_linesp4: the doubling step shared by two lane pairs, six multiplier passes
  // on the critical path instead of eleven) while its waves have a SIMD each (16,384 items = 1,024 waves: 0.75 ms against 1.1);
  // two lanes per item beyond (1.2 ms up to 32,768 items = 1,024 waves, 2.05 ms up to 65,536 = two waves per SIMD; the four-lane
  // kernel with two waves per SIMD takes 1.4 ms).  Measured, tools/dbg/r3_agg5.sh.
  const size_t lines4_max = lines4_env >= 0 ? (size_t)lines4_env : 16384;
  // the signature's pair apart?  Always when its lines come from a table; a general one only where it would cost a cliff
  const size_t G = last_mode ? mm - 1 : mm;
  const bool extra = last_mode == 1 || last_mode == 2 || (last_mode == 3 && G % 16384 == 0);
  const size_t M = extra ? G : mm;                                   // items of the chunks proper
  const size_t rem = M % round;
  const size_t left = M >= round && rem < 1024 ? rem : 0, body = M - left;
  const size_t nch = (body + round - 1) / round;
  auto chunk_cnt = [&](size_t k) { return k + 1 < nch ? round : body - k * round; };
  auto quarter = [](size_t q) { return (q + 3) / 4; };
  auto local_out = [&](size_t cnt) {
    size_t q = quarter(cnt);
    while (q > LOCAL) q = quarter(q);
    return q;
  };
  const size_t ql = left ? quarter(left) : 0, qx = extra ? 1 : 0;
  size_t Qmain = 0;
  for (size_t k = 0; k < nch; k++) Qmain += local_out(chunk_cnt(k));
  const size_t Qc = Qmain + ql + qx;
  const size_t q0max = quarter(chunk_cnt(0));
  // two ping-pong buffers: the first takes a chunk's quads and the first level over all chunks' values (fan-in >= 2), the second the levels after them
  const size_t half_qc = (Qc + 1) / 2;
  const size_t capA = E * (q0max > half_qc ? q0max : half_qc), capB = E * (quarter(q0max) > quarter(Qc) + 16 ? quarter(q0max) : quarter(Qc) + 16);
  const size_t lanes_max = row_stride(lanes_for(chunk_cnt(0))), lanes_left = left ? row_stride(lanes_for(left)) : 0, lanes_x = extra ? row_stride(lanes_for(1)) : 0;
  const size_t T_STRIDE = 128;
  if ((capA + E * Qc) * W1 >= ((size_t)1 << 30)) return 0;      // the fold kernels' 32-bit lane offsets (kernels.cuh wsu_ld_hfp6)
  const size_t words = E * LINE3_WORDS_H * (lanes_max + lanes_left + lanes_x) + (size_t)WS_F_WORDS * (capA + capB + E * Qc + T_STRIDE);
  if (lines_reserve(c, words * 4) != 0) return 0;
  uint32_t* lines3 = c->lines_ws;
  uint32_t* lines3_left = lines3 + E * LINE3_WORDS_H * lanes_max;
  uint32_t* lines3_x = lines3_left + E * LINE3_WORDS_H * lanes_left;
  uint32_t* bufA = lines3_x + E * LINE3_WORDS_H * lanes_x;
  uint32_t* bufB = bufA + (size_t)WS_F_WORDS * capA;
  uint32_t* comb = bufB + (size_t)WS_F_WORDS * capB;
  uint32_t* t68 = comb + (size_t)WS_F_WORDS * E * Qc;
  const size_t comb_stride = E * Qc;
  size_t comb_off = 0;
  for (size_t k = 0; k < nch; k++) {
    const size_t lo = k * round, cnt = chunk_cnt(k);
    const size_t nlanes = lanes_for(cnt), lanes = row_stride(nlanes);
    const uint32_t* pw = d_pairs + lo;
    const int32_t* bw = d_bad + lo;
    if (cnt <= lines4_max)
      KL(KID_LINESP4, k_linesp4, dim3(blocks_for(4 * cnt)), dim3(BLS_BLOCK), cnt, stride, pw, bw, lines3, lanes);
    else
      KL(KID_LINESP, k_linesp, dim3((unsigned)(nlanes / BLS_BLOCK)), dim3(BLS_BLOCK), cnt, cnt, stride, pw, bw, lines3, lines3, lanes, (size_t)0, cnt, 1);
    if (k == 0 && (left || extra)) {
      // the leftover items and the signature's pair: chunks of their own on the tail stream, their values straight into the
      // combined level.  They start when the first chunk's line kernel is through -- beside that kernel, whose waves fill the
      // machine's slots exactly, a few extra waves cost a whole chain's time (above) -- and their own chains hide beside the short
      // tasks of the chunk's product kernels
      if (!c->tail) {
        HIPCK(hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&c->ev_tail_fork, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&c->ev_tail_join, hipEventDisableTiming));
      }
      HIPCK(hipEventRecord(c->ev_tail_fork, c->stream));
      HIPCK(hipStreamWaitEvent(c->tail, c->ev_tail_fork, 0));
      if (extra) {
        const uint32_t* px = d_pairs + (mm - 1);
        const int32_t* bx = d_bad + (mm - 1);
        if (last_mode == 3) KLS(KID_TAIL, c->tail, k_linesp4, dim3(1), dim3(BLS_BLOCK), (size_t)1, stride, px, bx, lines3_x, lanes_x);
        else KLS(KID_TAIL, c->tail, k_lines_fixed, dim3(1), dim3(BLS_BLOCK), stride, px, bx, lines3_x, lanes_x, last_mode);
        KLS(KID_TAIL, c->tail, k_line_quad, dim3(1, (unsigned)E), dim3(BLS_BLOCK), (size_t)1, (size_t)1, bx, (const uint32_t*)lines3_x, lanes_x, comb, comb_stride, Qc,
                           Qmain + ql);
      }
      if (left) {
        KLS(KID_TAIL, c->tail, k_linesp4, dim3(blocks_for(4 * left)), dim3(BLS_BLOCK), left, stride, (const uint32_t*)(d_pairs + body),
                           (const int32_t*)(d_bad + body), lines3_left, lanes_left);
        KLS(KID_TAIL, c->tail, k_line_quad, dim3(blocks_for(2 * ql), (unsigned)E), dim3(BLS_BLOCK), left, ql, (const int32_t*)(d_bad + body),
                           (const uint32_t*)lines3_left, lanes_left, comb, comb_stride, Qc, Qmain);
      }
      const hipError_t e1 = hipGetLastError(), e2 = hipEventRecord(c->ev_tail_join, c->tail);
      if (e1 != hipSuccess || e2 != hipSuccess) {
        (void)hipStreamSynchronize(c->tail);
        return fail(BLSGPU_E_HIP, "tail-stream launch failed");
      }
    }
    size_t q = quarter(cnt);
    if (q <= LOCAL) {
      KL(KID_LINE_QUAD, k_line_quad, dim3(blocks_for(2 * q), (unsigned)E), dim3(BLS_BLOCK), cnt, q, bw, (const uint32_t*)lines3, lanes, comb, comb_stride, Qc, comb_off);
    } else {
      KL(KID_LINE_QUAD, k_line_quad, dim3(blocks_for(2 * q), (unsigned)E), dim3(BLS_BLOCK), cnt, q, bw, (const uint32_t*)lines3, lanes, bufA, E * q, q, (size_t)0);
      uint32_t *src = bufA, *dst = bufB;
      while (q > LOCAL) {
        const size_t qo = quarter(q);
        if (qo <= LOCAL)
          KL(KID_F12_FOLD4, k_f12_fold4, dim3(blocks_for(2 * qo), (unsigned)E), dim3(BLS_BLOCK), q, qo, 4, (const uint32_t*)src, E * q, q, comb, comb_stride, Qc, comb_off);
        else
          KL(KID_F12_FOLD4, k_f12_fold4, dim3(blocks_for(2 * qo), (unsigned)E), dim3(BLS_BLOCK), q, qo, 4, (const uint32_t*)src, E * q, q, dst, E * qo, qo, (size_t)0);
        uint32_t* t = src;
        src = dst;
        dst = t;
        q = qo;
      }
    }
    comb_off += q;
  }
  if (left || extra) HIPCK(hipStreamWaitEvent(c->stream, c->ev_tail_join, 0));
  // all chunks' values together
  {
    const uint32_t* src = comb;
    size_t q = Qc;
    uint32_t *dst = bufA, *other = bufB;
    // Levels that fill the machine on lane pairs (fan-in four, or five where that saves a level), down to at most ENGINE_FROM values
    // per entry; the rest on the engine, sixteen values per workgroup (two launches for up to 256 values): a product there takes
    // 3.7 us against 43 on a lane pair, and a lane-pair level of a few hundred tasks is nothing but its chain of fan - 1 products
    // (~0.13 ms).  1,025 values per entry -- a round of pairs and the signature's -- go 205, 52 on lane pairs, then 4, 1 on the engine.
    const long engine_from_env = knobs().tree_engine_from;
    const size_t ENGINE_FROM = engine_from_env >= 16 && engine_from_env <= 256 ? (size_t)engine_from_env : 128;
    int levels = 0;
    for (size_t cap = ENGINE_FROM; cap < q; cap *= 5) levels++;
    while (q > ENGINE_FROM) {
      size_t fan = 2, reach = ENGINE_FROM;
      for (int l = 1; l < levels; l++) reach *= 4;            // what the remaining levels can take at fan-in four
      while (fan < 5 && (q + fan - 1) / fan > reach) fan++;    // (five always suffices: `levels` was counted for fan-in five)
      levels--;
      const size_t qo = (q + fan - 1) / fan;
      KL(KID_F12_FOLD4, k_f12_fold4, dim3(blocks_for(2 * qo), (unsigned)E), dim3(BLS_BLOCK), q, qo, (int)fan, src, E * q, q, dst, E * qo, qo, (size_t)0);
      src = dst;
      uint32_t* t = dst;
      dst = other;
      other = t;
      q = qo;
    }
    if (q > 16) {
      const size_t qo = (q + 15) / 16;
      KL(KID_F12_TREE, k_f12_tree_seg, dim3((unsigned)qo, (unsigned)E), dim3(WIDE_ENGINE_BLOCK), q, src, E * q, q, dst, E * qo, qo);
      src = dst;
      q = qo;
    }
    KL(KID_F12_TREE, k_f12_tree_seg, dim3(1, (unsigned)E), dim3(WIDE_ENGINE_BLOCK), q, src, E * q, q, t68, T_STRIDE, (size_t)1);
    KL(KID_HORNER, k_f12_horner_wide, dim3(1), dim3(WIDE_ENGINE_BLOCK), (const uint32_t*)t68, T_STRIDE, d_f, stride);
  }
  HIPCK(hipGetLastError());
  *outputs = 1;
  *done = true;
  return 0;
}

// Miller loops of a pairing product over mm one-pair items (workspace stride given; flagged items contribute 1): leaves *outputs
// partial products at the start of the Fp12 workspace.  Round 3: two items share a merged line value (k_linesp, two passes),
// and one accumulator takes the values of `group` such pairs per step (k_millerfp).  The items are cut into chunks of 2 cnt
// items whose lane-pair counts fill whole workgroups -- and whole machine rounds (65,536 lane pairs) when there are that many:
// a single extra wave after a full round costs a lone wave's latency, ~7 ms -- and what is left (< 1,024 items: 262,145 pairs
// leave one) goes through k_miller1s on the context's tail stream BESIDE the chunks.  BLSGPU_MILLER_V1=1, fewer than 4,096
// items, or no memory for the line workspace: k_miller1s for everything.
int run_miller_product(Ctx* c, size_t mm, size_t stride, uint32_t* d_pairs, int32_t* d_bad, uint32_t* d_f, size_t* outputs, int last_mode = 0) {
  {
    bool done = false;
    if (int rc = run_miller_product_tree(c, mm, stride, d_pairs, d_bad, d_f, outputs, &done, last_mode)) return rc;
    if (done) return 0;
  }
  const size_t chunk_max = 2 * miller_chunk_items(), round = 65536;
  const size_t words_per_lane = (size_t)MILLER_ENTRIES * (LINE5_WORDS + LINE3_WORDS_H);
  auto nice = [&](size_t pairs) {          // virtual items of the next chunk when `pairs` virtual items could still be formed
    if (pairs > chunk_max) pairs = chunk_max;
    return pairs >= round ? pairs / round * round : pairs / 32 * 32;
  };
  if (mm >= 64 && mm <= round + 1023 && chunk_max) {
    // at most one machine round of lane pairs: one item per accumulator slot, plain lines (k_linesp pass 1, k_millerfp3); up to 1,023
    // items beyond the round go through k_miller1s on the tail stream beside them
    const size_t cnt = mm >= round ? round : mm, left = mm - cnt;      // below a round: every item gets a lane pair (the last workgroup may be partly idle)
    const size_t nlanes = lanes_for(cnt), lanes = row_stride(nlanes);
    if (lines_reserve(c, (size_t)MILLER_ENTRIES * LINE3_WORDS_H * 4 * lanes) == 0) {
      if (left) {
        if (!c->tail) {
          HIPCK(hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking));
          HIPCK(hipEventCreateWithFlags(&c->ev_tail_fork, hipEventDisableTiming));
          HIPCK(hipEventCreateWithFlags(&c->ev_tail_join, hipEventDisableTiming));
        }
        HIPCK(hipEventRecord(c->ev_tail_fork, c->stream));
        HIPCK(hipStreamWaitEvent(c->tail, c->ev_tail_fork, 0));
        KLS(KID_TAIL, c->tail, k_miller1s, dim3(blocks_for(2 * MILLER1_OUTPUTS(left))), dim3(BLS_BLOCK), left, stride, (const uint32_t*)(d_pairs + cnt),
                           (const int32_t*)(d_bad + cnt), d_f + cnt);
        const hipError_t e1 = hipGetLastError(), e2 = hipEventRecord(c->ev_tail_join, c->tail);
        if (e1 != hipSuccess || e2 != hipSuccess) {
          (void)hipStreamSynchronize(c->tail);
          return fail(BLSGPU_E_HIP, "tail-stream launch failed");
        }
      }
      const dim3 grid((unsigned)(nlanes / BLS_BLOCK));
      // k_linesp pass 1 on a "chunk" whose partners do not exist (half = cnt): every lane pair walks its own item and stores its lines
      KL(KID_LINESP, k_linesp, grid, dim3(BLS_BLOCK), cnt, cnt, stride, (const uint32_t*)d_pairs, (const int32_t*)d_bad, c->lines_ws, c->lines_ws, lanes, (size_t)0, cnt, 1);
      KL(KID_MILLERFP, k_millerfp3, grid, dim3(BLS_BLOCK), cnt, cnt, 1, (const int32_t*)d_bad, (const uint32_t*)c->lines_ws, lanes, d_f, stride, (size_t)0);
      if (left) HIPCK(hipStreamWaitEvent(c->stream, c->ev_tail_join, 0));
      HIPCK(hipGetLastError());
      *outputs = cnt + (left ? MILLER1_OUTPUTS(left) : 0);
      return 0;
    }
  }
  const size_t first_cnt = mm >= 4096 && chunk_max ? nice(mm / 2) : 0;
  if (first_cnt == 0 || lines_reserve(c, words_per_lane * 4 * row_stride(lanes_for(first_cnt))) != 0) {
    MILLER1_LAUNCH(mm, stride, d_pairs, d_bad, d_f);
    HIPCK(hipGetLastError());
    *outputs = MILLER1_OUTPUTS(mm);
    return 0;
  }
  // the plan: chunks [lo, lo + 2 cnt) until fewer than 1,024 items are left
  struct Chunk { size_t lo, cnt; };
  std::vector<Chunk> plan;
  size_t lo = 0;
  while (mm - lo >= 1024) {
    const size_t cnt = nice((mm - lo) / 2);
    plan.push_back({lo, cnt});
    lo += 2 * cnt;
  }
  size_t out0 = 0;
  for (const Chunk& ch : plan) {
    int group = (int)(ch.cnt / round);              // one machine round of accumulators when the chunk has several rounds of line values
    if (group < 1) group = 1;
    if (group > 4) group = 4;
    out0 += (ch.cnt + group - 1) / group;
  }
  const size_t left = mm - lo, left_out0 = out0;
  if (left) {                                       // first in time: a few waves whose latency hides under the chunks
    if (!c->tail) {
      HIPCK(hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking));
      HIPCK(hipEventCreateWithFlags(&c->ev_tail_fork, hipEventDisableTiming));
      HIPCK(hipEventCreateWithFlags(&c->ev_tail_join, hipEventDisableTiming));
    }
    HIPCK(hipEventRecord(c->ev_tail_fork, c->stream));
    HIPCK(hipStreamWaitEvent(c->tail, c->ev_tail_fork, 0));
    KLS(KID_TAIL, c->tail, k_miller1s, dim3(blocks_for(2 * MILLER1_OUTPUTS(left))), dim3(BLS_BLOCK), left, stride, (const uint32_t*)(d_pairs + lo),
                       (const int32_t*)(d_bad + lo), d_f + left_out0);
    const hipError_t e1 = hipGetLastError(), e2 = hipEventRecord(c->ev_tail_join, c->tail);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      (void)hipStreamSynchronize(c->tail);
      return fail(BLSGPU_E_HIP, "tail-stream launch failed");
    }
  }
  out0 = 0;
  for (const Chunk& ch : plan) {
    const size_t nlanes = lanes_for(ch.cnt), lanes = row_stride(nlanes);
    uint32_t* lines3 = c->lines_ws + (size_t)MILLER_ENTRIES * LINE5_WORDS * lanes;
    int group = (int)(ch.cnt / round);
    if (group < 1) group = 1;
    if (group > 4) group = 4;
    const size_t q = (ch.cnt + group - 1) / group;
    const dim3 grid((unsigned)(nlanes / BLS_BLOCK));
    // the chunk's items as a workspace of its own: item v pairs with item v + cnt
    const uint32_t* pw = d_pairs + ch.lo;
    const int32_t* bw = d_bad + ch.lo;
    KL(KID_LINESP, k_linesp, grid, dim3(BLS_BLOCK), 2 * ch.cnt, ch.cnt, stride, pw, bw, c->lines_ws, lines3, lanes, (size_t)0, ch.cnt, 1);
    KL(KID_LINESP, k_linesp, grid, dim3(BLS_BLOCK), 2 * ch.cnt, ch.cnt, stride, pw, bw, c->lines_ws, lines3, lanes, (size_t)0, ch.cnt, 2);
    KL(KID_MILLERFP, k_millerfp, dim3(blocks_for(2 * q)), dim3(BLS_BLOCK), ch.cnt, q, group, (const uint32_t*)c->lines_ws, lanes, d_f, stride, out0);
    out0 += q;
  }
  if (left) {
    HIPCK(hipStreamWaitEvent(c->stream, c->ev_tail_join, 0));
    out0 += MILLER1_OUTPUTS(left);
  }
  HIPCK(hipGetLastError());
  *outputs = out0;
  return 0;
}

// product of m Fp12 values in a workspace (stride given) folded into item 0
int run_f12_fold(Ctx* c, uint32_t* d_f, size_t m, size_t stride) {
  // halvings on lane pairs while there are thousands of values (parallel work, 0.15 ms per launch), then sixteen-way products on
  // the engine (k_f12_tree_wide, ~50 us per launch): 65,536 values are one value after 4 + 3 launches instead of 16
  const bool engine = wide_max_items() > 0 && coop_max_items() > 0;
  const size_t ENGINE_FROM = 4096;
  size_t cur = m;
  while (cur > (engine ? ENGINE_FROM : 1)) {
    size_t half = (cur + 1) / 2;
    KL(KID_F12_FOLD, k_f12_fold, dim3(blocks_for(half)), dim3(BLS_BLOCK), cur, half, d_f, stride);
    cur = half;
  }
  if (cur > 1) {
    const size_t cap = ENGINE_FROM / 16;                      // items per ping-pong buffer
    if (!c->fold_ws) HIPCK(hipMalloc((void**)&c->fold_ws, 2 * cap * (size_t)WS_F_WORDS * 4));
    const uint32_t* src = d_f;
    size_t sstride = stride;
    int which = 0;
    while (cur > 1) {
      const size_t outn = (cur + 15) / 16;
      uint32_t* dst = outn == 1 ? d_f : c->fold_ws + (size_t)which * cap * WS_F_WORDS;
      const size_t dstride = outn == 1 ? stride : cap;
      KL(KID_F12_FOLD, k_f12_tree_wide, dim3((unsigned)outn), dim3(WIDE_ENGINE_BLOCK), cur, src, sstride, dst, dstride);
      src = dst;
      sstride = dstride;
      which ^= 1;
      cur = outn;
    }
  }
  HIPCK(hipGetLastError());
  return 0;
}
// ... then the verdict of item 0
int run_f12_product_verdict(Ctx* c, uint32_t* d_f, size_t m, size_t stride, int32_t* d_verdict) {
  int rc = run_f12_fold(c, d_f, m, stride);
  if (rc) return rc;
  if (wide_max_items() > 0 && coop_max_items() > 0) KL(KID_FINALEXP_ONE, k_finalexp_wide_ws, dim3(1), dim3(WIDE_ENGINE_BLOCK), (const uint32_t*)d_f, stride, d_verdict);
  else if (coop_max_items() > 0) KL(KID_FINALEXP_ONE, k_finalexp_coop, dim3(1), dim3(BLS_BLOCK), d_f, stride, d_verdict);
  else KL(KID_FINALEXP_ONE, k_finalexp_ones, dim3(1), dim3(BLS_BLOCK), d_f, stride, d_verdict);
  HIPCK(hipGetLastError());
  return 0;
}

// lanes (lane pairs for G2) of the first stage of a point sum.  The cap leaves a few compute units' worth of wave slots free in
// every XCD: a grid that fills the machine EXACTLY (65,536 lane pairs = 2,048 waves = two per SIMD) takes twice as long as soon
// as other kernels hold a few slots, because some workgroups then wait for a whole round -- and the side streams do hold
// slots: the engine workgroups of the message hash and of the signature's Miller function take a whole CU each (their waves
// use 355 registers), and workgroups go round-robin to the 8 XCDs, so two of them in one XCD cost it 16 slots.  Measured
// (tools/dbg/acc.py, 2^20 keys): 1.15 / 1.17 ms at 58,368 / 57,344 lane pairs, 1.28 at 59,392, 1.48 at 61,440.
size_t accumulate_lanes(size_t n) {
  const long cap = knobs().acc_lanes;                     // default 57,344: 224 of 256 compute units at 256 lane pairs each
  size_t t = n / 4;
  if (t < 64) t = 64;
  if (t > (size_t)cap) t = (size_t)cap;
  return (t + 63) & ~(size_t)63;
}

bool msm_use_pippenger(size_t n);
bool msm_use_v1();
struct msm2_plan {
  int c, W, CH, Q, E;
  size_t nb, nchunks;
};
struct msm2_ws {
  msm2_plan p;
  uint32_t *aff, *cnt, *off, *cur, *idx, *tiles;
  uint64_t* subs;
  uint8_t *inf, *sums, *part;
  uint32_t *tab = nullptr, *jt = nullptr;     // weighted tables (k_msm2_tables) and their Jacobian staging: verify_secure only
};
int msm2_ws_take(Ctx* c, size_t n, int G, msm2_ws& w, bool tables = false);
size_t msm2_tables_bytes(size_t n, int G);
template <int G>
int run_msm2_prep(Ctx* c, const uint8_t* d_pts, int fmt, const uint32_t* d_perm, size_t n, msm2_ws& w);
template <int G>
int run_msm2_rest(Ctx* c, const uint8_t* d_scalars, size_t n, msm2_ws& w, uint8_t* d_out, bool normalize = true);
template <int G>
int run_msm_pippenger(Ctx* c, const uint8_t* d_pts, int fmt, const uint8_t* d_scalars, const uint32_t* d_perm, size_t n, uint8_t* d_out);

// m partial sums (RAW_PROJ, device) -> partials[0].  Wide levels of the tree run on the lane-local / lane-pair fold (one
// addition deep per launch, efficient while there are thousands of additions); the last 4,096 points go through the row-wide
// engine, sixteen per workgroup and launch (k_point_tree_wide), between the partials and a small second buffer.
const size_t POINT_TREE_START = 4096;
template <int G>
int run_point_fold(Ctx* c, uint8_t* d_partials, size_t m) {
  const size_t psz = G == 1 ? 144 : 288;
  size_t cur = m;
  const bool wide = wide_max_items() >= 1;               // BLSGPU_WIDE_MAX=0 switches the row-wide engine off everywhere
  while (cur > (wide ? POINT_TREE_START : 1)) {
    size_t half = (cur + 1) / 2;
    if (G == 2) KL(KID_POINT_FOLD, k_point_fold_g2s, dim3(blocks_for(2 * half)), dim3(BLS_BLOCK), cur, half, d_partials);
    else KL(KID_POINT_FOLD, k_point_fold<G>, dim3(blocks_for(half)), dim3(BLS_BLOCK), cur, half, d_partials);
    cur = half;
  }
  if (cur > 1) {
    uint8_t* d_tmp = (uint8_t*)arena_take(c, psz * ((POINT_TREE_START + WIDE_PT_POINTS - 1) / WIDE_PT_POINTS));
    if (!d_tmp) return fail(BLSGPU_E_HIP, "internal: arena too small");
    const uint8_t* src = d_partials;
    while (cur > 1) {
      const size_t nb = (cur + WIDE_PT_POINTS - 1) / WIDE_PT_POINTS;
      uint8_t* dst = (nb == 1 || src != d_partials) ? d_partials : d_tmp;   // one workgroup reads all its inputs before it writes
      KL(KID_POINT_FOLD, k_point_tree_wide<G>, dim3((unsigned)nb), dim3(WIDE_ENGINE_BLOCK), cur, src, dst);
      src = dst;
      cur = nb;
    }
  }
  HIPCK(hipGetLastError());
  return 0;
}

// sum (or sum of scalar multiples) of n points of group G into partials[0] (RAW_PROJ, device)
template <int G>
int run_point_sum(Ctx* c, const uint8_t* d_pts, int fmt, const uint8_t* d_scalars, const uint32_t* d_perm, size_t n,
                  uint8_t* d_partials, size_t T) {
  if (d_scalars && msm_use_pippenger(n)) {
    if (msm_use_v1()) return run_msm_pippenger<G>(c, d_pts, fmt, d_scalars, d_perm, n, d_partials);
    msm2_ws w;
    int rc = msm2_ws_take(c, n, G, w);
    if (rc) return rc;
    if ((rc = run_msm2_prep<G>(c, d_pts, fmt, d_perm, n, w))) return rc;
    return run_msm2_rest<G>(c, d_scalars, n, w, d_partials);
  }
  unsigned nb = blocks_for(T);
  if (d_scalars)
    KL(KID_ACCUM, (k_accumulate<G, 1>), dim3(nb), dim3(BLS_BLOCK), n, d_pts, fmt, d_scalars, d_perm, d_partials, T);
  else
    if (G == 2 && !d_perm) KL(KID_ACCUM, k_accumulate_g2s, dim3(blocks_for(2 * T)), dim3(BLS_BLOCK), n, d_pts, fmt, d_partials, T);
    else KL(KID_ACCUM, (k_accumulate<G, 0>), dim3(nb), dim3(BLS_BLOCK), n, d_pts, fmt, d_scalars, d_perm, d_partials, T);
  return run_point_fold<G>(c, d_partials, T);
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
  if (knobs().msm_c) p.c = (int)knobs().msm_c;         // tuning overrides (window bits, buckets per chunk lane),
  if (knobs().msm_ch) p.CH = (int)knobs().msm_ch;      // clamped to what the kernels assume
  if (p.c < 4) p.c = 4;
  if (p.c > 16) p.c = 16;
  if (p.CH < 1) p.CH = 1;
  while (p.CH & (p.CH - 1)) p.CH &= p.CH - 1;                       // a power of two, so that it divides 2^c and 2^clast
  if (p.CH > (1 << p.c)) p.CH = 1 << p.c;
  p.W = 255 / p.c;                      // scalars are < r < 2^255
  p.clast = 255 - p.c * (p.W - 1);      // the last window takes the remainder: c <= clast < 2c
  p.nb = ((size_t)(p.W - 1) << p.c) + ((size_t)1 << p.clast);
  p.nchunks = p.nb / p.CH;
  return p;
}
size_t msm2_ws_bytes(size_t n);
size_t msm_ws_bytes(size_t n) {         // callers reserve for either group and either generation
  size_t need = msm2_ws_bytes(n);
  for (int G = 1; G <= 2; G++) {
    msm_plan p = msm_make_plan(n, G);
    size_t b = 3 * pad256(4 * p.nb) + pad256(4 * n * p.W) + pad256(4 * scan_tiles(p.nb)) + pad256(288 * p.nb) + pad256(288 * p.nchunks) + 4096;
    if (b > need) need = b;
  }
  return need;
}
// ---- second-generation MSM (csrc/msm2.cuh): endomorphism split, signed digits, mixed additions --------------------
msm2_plan msm2_make_plan(size_t n, int G) {
  msm2_plan p;
  const int bits = G == 1 ? 128 : 64;     // width of the sub-scalars (k = a0 + a1 z^2 on G1, base-z digits on G2)
  p.E = G == 1 ? 2 : 4;
  int lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  p.c = lg - 3;                           // target window width: 13 bits at 65,536 points (measured sweep, tools/dbg/msm.py)
  if (p.c < 6) p.c = 6;
  if (p.c > 14) p.c = 14;
  p.CH = 4;                               // short chunks: the chunk lanes are a latency chain (4 / 8 / 16 measured: 1.1 / 1.4 / 2.1 ms)
  if (knobs().msm2_c) p.c = (int)knobs().msm2_c;      // tuning overrides, clamped to what the kernels assume
  if (knobs().msm2_ch) p.CH = (int)knobs().msm2_ch;
  if (p.c < 4) p.c = 4;
  if (p.c > 16) p.c = 16;
  // bits + 1 positions (one spare for the recoding carry) in W windows whose widths differ by at most one (msm2.cuh)
  p.W = (bits + 1 + p.c / 2) / p.c;
  if (p.W < 1) p.W = 1;
  const int base = (bits + 1) / p.W, rem = (bits + 1) % p.W;
  if (p.CH < 1) p.CH = 1;
  while (p.CH & (p.CH - 1)) p.CH &= p.CH - 1;
  if (p.CH > (1 << (base - 1))) p.CH = 1 << (base - 1);             // a chunk never straddles two windows
  p.nb = ((size_t)rem << base) + ((size_t)(p.W - rem) << (base - 1));
  p.nchunks = p.nb / p.CH;
  // parts per bucket: about 130,000 (bucket, part) lanes keep every SIMD busy through the uneven bucket sizes (measured:
  // 1 / 2 / 4 parts at 40,960 bucket lanes: 2.2 / 1.3 / 0.8 ms for G1); a fully parallel pass merges the parts afterwards
  const size_t lanes = p.nb * (G == 1 ? 1 : 2);
  p.Q = (int)((131072 + lanes - 1) / lanes);
  if (knobs().msm2_q) p.Q = (int)knobs().msm2_q;
  if (p.Q < 1) p.Q = 1;
  if (p.Q > 8) p.Q = 8;
  return p;
}
size_t msm2_ws_bytes(size_t n) {          // callers reserve for either group
  size_t need = 0;
  for (int G = 1; G <= 2; G++) {
    const msm2_plan p = msm2_make_plan(n, G);
    const size_t affw = (G == 1 ? 2 : 4) * FP_NL;
    size_t b = pad256(4 * affw * p.E * n) + pad256(n) + pad256(32 * n) + 3 * pad256(4 * p.nb) + pad256(4 * n * p.E * p.W) + pad256(4 * scan_tiles(p.nb)) +
               pad256(288 * p.nb * p.Q) + pad256(288 * p.nchunks) + 4096;
    if (b > need) need = b;
  }
  return need;
}
// the weighted tables of k_msm2_tables: W affine entries per image, and 4 coordinates per window of Jacobian staging
size_t msm2_tables_bytes(size_t n, int G) {
  const msm2_plan p = msm2_make_plan(n, G);
  const size_t affw = (G == 1 ? 2 : 4) * FP_NL;
  return pad256(4 * affw * p.E * p.W * n) + pad256(4 * (affw / 2) * 4 * (p.W > 1 ? p.W - 1 : 1) * n) + 512;
}
int msm2_ws_take(Ctx* c, size_t n, int G, msm2_ws& w, bool tables) {
  w.p = msm2_make_plan(n, G);
  const msm2_plan& p = w.p;
  const size_t affw = (G == 1 ? 2 : 4) * FP_NL;
  if (tables) {
    w.tab = (uint32_t*)arena_take(c, 4 * affw * p.E * p.W * n);
    w.jt = (uint32_t*)arena_take(c, 4 * (affw / 2) * 4 * (p.W > 1 ? p.W - 1 : 1) * n);
    if (!w.tab || !w.jt) return fail(BLSGPU_E_HIP, "internal: arena too small");
  }
  w.aff = (uint32_t*)arena_take(c, 4 * affw * p.E * n);
  w.inf = (uint8_t*)arena_take(c, n);
  w.subs = (uint64_t*)arena_take(c, 32 * n);
  w.cnt = (uint32_t*)arena_take(c, 4 * p.nb);
  w.off = (uint32_t*)arena_take(c, 4 * p.nb);
  w.cur = (uint32_t*)arena_take(c, 4 * p.nb);
  w.idx = (uint32_t*)arena_take(c, 4 * n * p.E * p.W);
  w.tiles = (uint32_t*)arena_take(c, 4 * scan_tiles(p.nb));
  w.sums = (uint8_t*)arena_take(c, 288 * p.nb * p.Q);
  w.part = (uint8_t*)arena_take(c, 288 * p.nchunks);
  if (!w.aff || !w.inf || !w.subs || !w.cnt || !w.off || !w.cur || !w.idx || !w.tiles || !w.sums || !w.part)
    return fail(BLSGPU_E_HIP, "internal: arena too small");
  return 0;
}
// the scalar-independent part: every point to affine, with its endomorphism images
template <int G>
int run_msm2_prep(Ctx* c, const uint8_t* d_pts, int fmt, const uint32_t* d_perm, size_t n, msm2_ws& w) {
  KL(KID_MSM_PREP, k_msm2_prep<G>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_pts, fmt, d_perm, w.aff, w.inf);
  if (w.tab)     // every window's multiple of every image, while the caller's host still hashes (verify_secure)
    KL(KID_MSM_PREP, k_msm2_tables<G>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint32_t*)w.aff, (const uint8_t*)w.inf, w.p.W, w.tab, w.jt);
  HIPCK(hipGetLastError());
  return 0;
}
// ... and the rest: digits, bucket lists, bucket sums, window weights, fold.  d_out[0] = the sum (RAW_PROJ, Z = 1)
template <int G>
int run_msm2_rest(Ctx* c, const uint8_t* d_scalars, size_t n, msm2_ws& w, uint8_t* d_out, bool normalize) {
  const msm2_plan& p = w.p;
  HIPCK(hipMemsetAsync(w.cnt, 0, 4 * p.nb, c->stream));
  HIPCK(hipMemsetAsync(w.cur, 0, 4 * p.nb, c->stream));
  KL(KID_MSM_SORT, k_msm2_count<G>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_scalars, (const uint8_t*)w.inf, p.W, w.cnt, w.subs);
  int rc = run_scan_u32(c, KID_MSM_SORT, p.nb, w.cnt, w.off, w.tiles);
  if (rc) return rc;
  KL(KID_MSM_SORT, k_msm2_fill<G>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint64_t*)w.subs, (const uint8_t*)w.inf, p.W, (const uint32_t*)w.off, w.cur, w.idx);
  const size_t nbq = p.nb * p.Q;
  const uint32_t* pts_ws = w.tab ? w.tab : w.aff;
  const int tabW = w.tab ? p.W : 0, weighted = w.tab ? 1 : 0;
  if (G == 2) KL(KID_MSM_BUCKET, k_msm2_bucket_g2s, dim3(blocks_for(2 * nbq)), dim3(BLS_BLOCK), p.nb, p.Q, pts_ws, (const uint32_t*)w.cnt, (const uint32_t*)w.off, (const uint32_t*)w.idx, w.sums, tabW);
  else KL(KID_MSM_BUCKET, k_msm2_bucket_g1, dim3(blocks_for(nbq)), dim3(BLS_BLOCK), p.nb, p.Q, pts_ws, (const uint32_t*)w.cnt, (const uint32_t*)w.off, (const uint32_t*)w.idx, w.sums, tabW);
  if (p.Q > 1) {
    if (G == 2) KL(KID_MSM_MERGE, k_msm2_merge_g2s, dim3(blocks_for(2 * p.nb)), dim3(BLS_BLOCK), p.nb, p.Q, w.sums);
    else KL(KID_MSM_MERGE, k_msm2_merge_g1, dim3(blocks_for(p.nb)), dim3(BLS_BLOCK), p.nb, p.Q, w.sums);
  }
  if (G == 2) KL(KID_MSM_CHUNK, k_msm2_chunk_g2q, dim3(blocks_for(4 * p.nchunks)), dim3(BLS_BLOCK), p.W, p.CH, p.Q, (const uint8_t*)w.sums, w.part, weighted);
  else KL(KID_MSM_CHUNK, k_msm2_chunk_g1p, dim3(blocks_for(2 * p.nchunks)), dim3(BLS_BLOCK), p.W, p.CH, p.Q, (const uint8_t*)w.sums, w.part, weighted);
  if (int rc = run_point_fold<G>(c, w.part, p.nchunks)) return rc;
  // Z = 1 makes the output bytes independent of the (atomic) bucket fill order; a caller that only feeds the point to the
  // pairing stages skips it
  if (normalize) KL(KID_MSM_NORM, k_normalize<G>, dim3(1), dim3(BLS_BLOCK), w.part);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(d_out, w.part, G == 1 ? 144 : 288, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
bool msm_use_v1() { return knobs().msm_v1 != 0; }   // A/B switch: the first-generation bucket method (unsigned digits, full additions)

bool msm_use_pippenger(size_t n) { return !knobs().msm_naive && n >= 1024; }
template <int G>
int run_msm_pippenger(Ctx* c, const uint8_t* d_pts, int fmt, const uint8_t* d_scalars, const uint32_t* d_perm, size_t n, uint8_t* d_out) {
  msm_plan p = msm_make_plan(n, G);
  uint32_t* d_cnt = (uint32_t*)arena_take(c, 4 * p.nb);
  uint32_t* d_off = (uint32_t*)arena_take(c, 4 * p.nb);
  uint32_t* d_cur = (uint32_t*)arena_take(c, 4 * p.nb);
  uint32_t* d_idx = (uint32_t*)arena_take(c, 4 * n * p.W);
  uint32_t* d_tiles = (uint32_t*)arena_take(c, 4 * scan_tiles(p.nb));
  uint8_t* d_sums = (uint8_t*)arena_take(c, 288 * p.nb);
  uint8_t* d_part = (uint8_t*)arena_take(c, 288 * p.nchunks);
  if (!d_cnt || !d_off || !d_cur || !d_idx || !d_tiles || !d_sums || !d_part) return fail(BLSGPU_E_HIP, "internal: arena too small");
  HIPCK(hipMemsetAsync(d_cnt, 0, 4 * p.nb, c->stream));
  HIPCK(hipMemsetAsync(d_cur, 0, 4 * p.nb, c->stream));
  KL(KID_MSM_SORT, k_msm_count, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_scalars, p.c, p.W, p.clast, d_cnt);
  {
    int rc = run_scan_u32(c, KID_MSM_SORT, p.nb, d_cnt, d_off, d_tiles);
    if (rc) return rc;
  }
  KL(KID_MSM_SORT, k_msm_fill, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_scalars, p.c, p.W, p.clast, d_off, d_cur, d_idx);
  if (G == 2) KL(KID_MSM_BUCKET, k_msm_bucket_g2s, dim3(blocks_for(2 * p.nb)), dim3(BLS_BLOCK), p.nb, d_pts, fmt, d_perm, d_cnt, d_off, d_idx, d_sums);
  else KL(KID_MSM_BUCKET, k_msm_bucket<G>, dim3(blocks_for(p.nb)), dim3(BLS_BLOCK), p.nb, d_pts, fmt, d_perm, d_cnt, d_off, d_idx, d_sums);
  if (G == 2) KL(KID_MSM_CHUNK, k_msm_chunk_g2q, dim3(blocks_for(4 * p.nchunks)), dim3(BLS_BLOCK), p.c, p.W, p.clast, p.CH, d_sums, d_part);
  else KL(KID_MSM_CHUNK, k_msm_chunk_g1p, dim3(blocks_for(2 * p.nchunks)), dim3(BLS_BLOCK), p.c, p.W, p.clast, p.CH, d_sums, d_part);
  if (int rc = run_point_fold<G>(c, d_part, p.nchunks)) return rc;
  KL(KID_MSM_NORM, k_normalize<G>, dim3(1), dim3(BLS_BLOCK), d_part);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(d_out, d_part, G == 1 ? 144 : 288, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}


// ---- hash_public_keys_with_sorted on the device (reference src/secure_aggregation.rs:41-103,269-335) -----------------
// Workspace of the key sort / coefficient step, carved from the arena.
struct keysort_ws {
  uint32_t *perm_a, *perm_b, *hist, *tiles, *flag;
  uint8_t* sorted;
  size_t ntiles;
};
size_t keysort_ws_bytes(size_t n, size_t width) {
  const size_t nt = (n + UK_SORT_TILE - 1) / UK_SORT_TILE;
  return 2 * pad256(4 * n) + pad256(4 * 256 * nt) + pad256(4 * scan_tiles(256 * nt)) + pad256(width * n) + 5 * 256 + 64;
}
int keysort_ws_take(Ctx* c, size_t n, size_t width, keysort_ws& w) {
  w.ntiles = (n + UK_SORT_TILE - 1) / UK_SORT_TILE;
  w.perm_a = (uint32_t*)arena_take(c, 4 * n);
  w.perm_b = (uint32_t*)arena_take(c, 4 * n);
  w.hist = (uint32_t*)arena_take(c, 4 * 256 * w.ntiles);
  w.tiles = (uint32_t*)arena_take(c, 4 * scan_tiles(256 * w.ntiles));
  w.sorted = (uint8_t*)arena_take(c, width * n);
  w.flag = (uint32_t*)arena_take(c, 64);
  if (!w.perm_a || !w.perm_b || !w.hist || !w.tiles || !w.sorted || !w.flag) return fail(BLSGPU_E_HIP, "internal: arena too small");
  return 0;
}
// stable LSD radix sort on key bytes [0, nbytes): the result lands in w.perm_a
int run_key_sort_passes(Ctx* c, const uint8_t* d_kb, size_t n, size_t width, int nbytes, keysort_ws& w) {
  // nbytes passes ping-pong between the two buffers; start so that the last pass writes perm_a
  uint32_t* bufs[2] = {w.perm_a, w.perm_b};
  int dst = (nbytes & 1) ? 0 : 1;
  const uint32_t* src = nullptr;   // identity
  const unsigned nt = (unsigned)w.ntiles;
  for (int b = nbytes - 1; b >= 0; b--) {
    KL(KID_KEY_SORT, k_rs_hist, dim3(nt), dim3(BLS_BLOCK), n, d_kb, width, b, src, w.hist, w.ntiles);
    int rc = run_scan_u32(c, KID_KEY_SORT, 256 * w.ntiles, w.hist, w.hist, w.tiles);
    if (rc) return rc;
    KL(KID_KEY_SORT, k_rs_scatter, dim3(nt), dim3(BLS_BLOCK), n, d_kb, width, b, src, (const uint32_t*)w.hist, w.ntiles, bufs[dst]);
    src = bufs[dst];
    dst ^= 1;
  }
  HIPCK(hipGetLastError());
  return 0;
}
// The keys are compressed curve points, i.e. their leading bytes are as good as random (the first byte carries ~5 bits: the
// flag bits and the top of a 381-bit coordinate): sort on a prefix only and check that no two neighbours tie there; only if
// they do (duplicate keys, adversarial prefixes) sort again on every byte.  A pass per byte, so the prefix is as short as
// keeps a tie among honest keys below 2^-10 per call: 2 log2(n) + 13 bits -- six bytes at 65,536 keys (eight cost 0.17 ms more).
// Leaves the permutation in w.perm_a, the sorted concatenation in the pinned host buffer (c->hpin) and synchronises.
int keysort_prefix(size_t n) {
  int lg = 0;
  while (((size_t)1 << lg) < n) lg++;
  int b = (2 * lg + 13 + 7) / 8;
  if (b < 4) b = 4;
  if (b > 8) b = 8;
  return b;
}
int run_key_sort_to_host(Ctx* c, const uint8_t* d_kb, size_t n, size_t width, keysort_ws& w, hipEvent_t ev_ready, bool* used_full) {
  int rc = pinned_reserve(c, width * n + 64);
  if (rc) return rc;
  uint32_t* h_flag = (uint32_t*)(c->hpin + ((width * n + 63) & ~(size_t)63));
  for (int attempt = 0; attempt < 2; attempt++) {
    const int nbytes = attempt == 0 ? keysort_prefix(n) : (int)width;
    if ((rc = run_key_sort_passes(c, d_kb, n, width, nbytes, w))) return rc;
    HIPCK(hipMemsetAsync(w.flag, 0, 4, c->stream));
    if (attempt == 0)
      KL(KID_KEY_SORT, k_keys_tie_flag, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_kb, width, (size_t)keysort_prefix(n), (const uint32_t*)w.perm_a, w.flag);
    KL(KID_KEY_SORT, k_keys_gather, dim3(blocks_for(n * (width / 4))), dim3(BLS_BLOCK), n, d_kb, width, (const uint32_t*)w.perm_a, w.sorted);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(c->hpin, w.sorted, width * n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipMemcpyAsync(h_flag, w.flag, 4, hipMemcpyDeviceToHost, c->stream));
    if (ev_ready && attempt == 0) {
      HIPCK(hipEventRecord(ev_ready, c->stream));
      return 0;                       // the caller enqueues more work, then calls run_key_sort_finish
    }
    HIPCK(hipStreamSynchronize(c->stream));
    if (*h_flag == 0) break;
    if (used_full) *used_full = true;
  }
  return 0;
}
// second half of the overlapped form: wait for the prefix-sorted bytes; redo with the full sort when the prefix tied
int run_key_sort_finish(Ctx* c, const uint8_t* d_kb, size_t n, size_t width, keysort_ws& w, hipEvent_t ev_ready, bool* used_full) {
  HIPCK(hipEventSynchronize(ev_ready));
  const uint32_t* h_flag = (const uint32_t*)(c->hpin + ((width * n + 63) & ~(size_t)63));
  if (*h_flag == 0) return 0;
  if (used_full) *used_full = true;
  int rc = run_key_sort_passes(c, d_kb, n, width, (int)width, w);
  if (rc) return rc;
  KL(KID_KEY_SORT, k_keys_gather, dim3(blocks_for(n * (width / 4))), dim3(BLS_BLOCK), n, d_kb, width, (const uint32_t*)w.perm_a, w.sorted);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(c->hpin, w.sorted, width * n, hipMemcpyDeviceToHost, c->stream));
  HIPCK(hipStreamSynchronize(c->stream));
  return 0;
}
// H = SHA-256(sorted key bytes): ONE sequential hash stream (reference :45-59), which is why it runs on a host core
// (SHA-NI: ~2 GB/s; a GPU lane would need ~100 ms for it)
void keys_digest_host(const uint8_t* sorted, size_t bytes, uint8_t H[32]) {
  host_sha256 h;
  h.update(sorted, bytes);
  h.final(H);
}
// t_p for every sorted position p on the device.  sorted_order: d_scal[p]; else d_scal[perm[p] - base] for keys of the shard
int run_coefficients(Ctx* c, const uint8_t* d_H, const uint32_t* d_perm, size_t n, size_t base, size_t count, int sorted_order,
                     uint8_t* d_scal, int32_t* d_zero_flag) {
  KL(KID_COEFF, k_sha256_coeff, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_H, d_perm, base, count, sorted_order, d_scal, d_zero_flag);
  HIPCK(hipGetLastError());
  return 0;
}

// ---- Basic's duplicate-message rule on device-resident messages (util_kernels.cuh) -> d_out2 = (old, i) or (~0, ~0)
size_t dup_ws_bytes(size_t n) {
  size_t cap = 64;
  while (cap < 2 * n) cap <<= 1;
  return 2 * pad256(4 * cap) + pad256(4 * n) + 1024;
}
int run_first_duplicate(Ctx* c, const uint8_t* d_msgs, const uint64_t* d_offs, size_t n, uint64_t* d_out2) {
  if (n >= 0xffffffffull) return fail(BLSGPU_E_ARG, "more than 2^32 - 2 messages");
  size_t cap = 64;
  while (cap < 2 * n) cap <<= 1;
  uint32_t* tab = (uint32_t*)arena_take(c, 4 * cap);
  uint32_t* minidx = (uint32_t*)arena_take(c, 4 * cap);
  uint32_t* slot_of = (uint32_t*)arena_take(c, 4 * (n ? n : 1));
  uint32_t* best = (uint32_t*)arena_take(c, 64);
  if (!tab || !minidx || !slot_of || !best) return fail(BLSGPU_E_HIP, "internal: arena too small");
  HIPCK(hipMemsetAsync(tab, 0xff, 4 * cap, c->stream));
  HIPCK(hipMemsetAsync(minidx, 0xff, 4 * cap, c->stream));
  HIPCK(hipMemsetAsync(best, 0xff, 4, c->stream));
  if (n) {
    KL(KID_DUP, k_dup_insert, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, d_msgs, d_offs, (uint32_t)(cap - 1), tab, minidx, slot_of);
    KL(KID_DUP, k_dup_find, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint32_t*)slot_of, (const uint32_t*)minidx, best);
  }
  KL(KID_DUP, k_dup_fin, dim3(1), dim3(BLS_BLOCK), (const uint32_t*)best, (const uint32_t*)slot_of, (const uint32_t*)minidx, d_out2);
  HIPCK(hipGetLastError());
  return 0;
}

}  // namespace

// =========================================================================================================
extern "C" {

// binds one device: a pool of contexts on it
static int bind_device(int device, int ndev) {
  if (device >= ndev) return fail(BLSGPU_E_ARG, "device ordinal out of range");
  HIPCK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(BLSGPU_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
  const int nctx = (int)knobs().contexts;   // contexts per device (BLSGPU_CONTEXTS, default 2, 1..16): concurrent callers beyond this queue up
  Device* d = new Device();
  d->dev = device;
  for (int k = 0; k < nctx; k++) {
    Ctx* c = new Ctx();
    c->dev = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
      e = hipHostMalloc((void**)&c->hsmall, HSMALL_BYTES, hipHostMallocDefault);
      if (e != hipSuccess) (void)hipStreamDestroy(c->stream);
    }
    if (e == hipSuccess) {
      // the hand-over flags of the single-verdict kernels (stream_epoch_next): here, not on first use -- an allocation inside a
      // call costs that call a few hundred microseconds
      const size_t fb = (size_t)WPOST2_MAX_ITEMS * WSTREAM_FLAGS * 4;
      e = hipMalloc((void**)&c->stream_flags, fb);
      if (e == hipSuccess) e = hipMemset(c->stream_flags, 0, fb);
      if (e == hipSuccess) c->stream_epoch = 1;         // cleared: the first launch publishes with 2
      if (e != hipSuccess) {
        if (c->stream_flags) (void)hipFree(c->stream_flags);
        (void)hipHostFree(c->hsmall);
        (void)hipStreamDestroy(c->stream);
      }
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      delete c;
      for (Ctx* q : d->pool) {
        (void)hipStreamDestroy(q->stream);
        (void)hipHostFree(q->hsmall);
        (void)hipFree(q->stream_flags);
        delete q;
      }
      delete d;
      return fail(BLSGPU_E_HIP, std::string("creating a context (stream / pinned memory): ") + hipGetErrorString(e));
    }
    d->pool.push_back(c);
  }
  g_devices.push_back(d);
  return 0;
}

static void release_devices() {
  for (Device* d : g_devices) {
    (void)hipSetDevice(d->dev);
    for (Ctx* c : d->pool) {
      std::lock_guard<std::mutex> lk(c->mu);   // waits for a call in flight on this context
      (void)hipStreamSynchronize(c->stream);
      if (c->arena) (void)hipFree(c->arena);
      if (c->lines_ws) (void)hipFree(c->lines_ws);
      if (c->fold_ws) (void)hipFree(c->fold_ws);
      if (c->stream_flags) (void)hipFree(c->stream_flags);
      if (c->ev_tail_fork) (void)hipEventDestroy(c->ev_tail_fork);
      if (c->ev_tail_join) (void)hipEventDestroy(c->ev_tail_join);
      if (c->tail) (void)hipStreamDestroy(c->tail);
      if (c->ev_host) (void)hipEventDestroy(c->ev_host);
      if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
      if (c->ev_join) (void)hipEventDestroy(c->ev_join);
      if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
      if (c->side2) (void)hipStreamDestroy(c->side2);
      if (c->side) (void)hipStreamDestroy(c->side);
      if (c->hpin) (void)hipHostFree(c->hpin);
      if (c->hsmall) (void)hipHostFree(c->hsmall);
      for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
      (void)hipStreamDestroy(c->stream);
    }
  }
  for (Device* d : g_devices) {
    for (Ctx* c : d->pool) delete c;
    delete d;
  }
  g_devices.clear();
}

// BLSGPU_STRICT_ENV=1: refuse to start over an environment the knob table does not fully understand (before any device is touched)
static int strict_env_check() {
  if (!initialised()) g_knobs = parse_knobs();
  const Knobs& k = knobs();
  if (k.strict_env && !k.problem.empty()) return fail(BLSGPU_E_ARG, "BLSGPU_STRICT_ENV: " + k.problem);
  return 0;
}
int blsgpu_init(int device) try {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (int rc = strict_env_check()) return rc;
  if (initialised()) {
    if (device >= 0 && device != g_devices[0]->dev)
      return fail(BLSGPU_E_ARG, "blsgpu_init: the library is already bound to device " + std::to_string(g_devices[0]->dev) +
                                    " (call blsgpu_shutdown first to rebind)");
    return 0;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return fail(BLSGPU_E_NO_DEVICE, "no HIP device available: libblsgpu has no CPU fallback");
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  return bind_device(device, ndev);
}
API_CATCH

/* One process driving several GPUs of the node (the in-library alternative to one process per GPU): binds the
 * first ndev visible devices (ndev <= 0: all of them).  The four verify entry points then shard their items over the
 * bound devices (contiguous ranges, one host thread per device) and fold the per-device partial results -- key sums,
 * Fp12 Miller products, MSM partials -- on device 0, exactly as the one-process-per-GPU path does over RCCL.
 * BLSGPU_FAKE_DEVICES=k (testing on a one-GPU box) binds k logical devices that all map to the one physical GPU. */
int blsgpu_init_devices(int ndev_req) try {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (int rc = strict_env_check()) return rc;
  if (initialised()) {
    if (ndev_req > 0 && (size_t)ndev_req != g_devices.size())
      return fail(BLSGPU_E_ARG, "blsgpu_init_devices: the library is already bound to " + std::to_string(g_devices.size()) +
                                    " device(s) (call blsgpu_shutdown first to rebind)");
    return (int)g_devices.size();
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return fail(BLSGPU_E_NO_DEVICE, "no HIP device available: libblsgpu has no CPU fallback");
  }
  const int fake = (int)knobs().fake_devices;
  int want = ndev_req > 0 ? ndev_req : (fake > 0 ? fake : ndev);
  if (fake <= 0 && want > ndev) return fail(BLSGPU_E_ARG, "blsgpu_init_devices: more devices requested than visible");
  if (want > 64) return fail(BLSGPU_E_ARG, "blsgpu_init_devices: at most 64 devices");
  for (int k = 0; k < want; k++) {
    int rc = bind_device(fake > 0 ? k % ndev : k, ndev);
    if (rc) {
      release_devices();
      return rc;
    }
  }
  // peer access so that a device can read a shard that lives on another device's memory (xGMI loads); failures are not
  // fatal: shards of host buffers never need it
  g_peer_ok = true;
  for (size_t a = 0; a < g_devices.size(); a++)
    for (size_t b = 0; b < g_devices.size(); b++) {
      if (g_devices[a]->dev == g_devices[b]->dev) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g_devices[a]->dev, g_devices[b]->dev) == hipSuccess && can) {
        (void)hipSetDevice(g_devices[a]->dev);
        hipError_t pe = hipDeviceEnablePeerAccess(g_devices[b]->dev, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) g_peer_ok = false;
        if (pe != hipSuccess) (void)hipGetLastError();
      } else {
        g_peer_ok = false;
      }
    }
  (void)hipSetDevice(g_devices[0]->dev);
  return (int)g_devices.size();
}
API_CATCH

int blsgpu_device_count(void) { return (int)g_devices.size(); }

void blsgpu_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (!initialised()) return;
  release_devices();
}

int blsgpu_profile_enable(int on) try {
  if (!initialised()) return NOT_INIT();
  for (Device* d : g_devices)
    for (Ctx* c : d->pool) {
      std::lock_guard<std::mutex> lk(c->mu);
      c->prof_on = on != 0;
      for (int k = 0; k < KID_COUNT; k++) {
        c->prof_ms[k] = 0;
        c->prof_cnt[k] = 0;
      }
    }
  return 0;
}
API_CATCH
int blsgpu_profile_count(void) { return KID_COUNT; }
int blsgpu_profile_get(int kernel_id, char* name, size_t name_cap, double* total_ms, uint64_t* launches) try {
  if (!initialised()) return NOT_INIT();
  if (kernel_id < 0 || kernel_id >= KID_COUNT) return fail(BLSGPU_E_ARG, "kernel id out of range");
  if (name && name_cap) {
    strncpy(name, KID_NAMES[kernel_id], name_cap - 1);
    name[name_cap - 1] = 0;
  }
  double ms = 0;
  uint64_t cnt = 0;
  for (Device* d : g_devices)
    for (Ctx* c : d->pool) {
      std::lock_guard<std::mutex> lk(c->mu);
      ms += c->prof_ms[kernel_id];
      cnt += c->prof_cnt[kernel_id];
    }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = cnt;
  return 0;
}
API_CATCH

size_t blsgpu_last_error(char* buf, size_t cap) {
  if (buf && cap) {
    size_t k = std::min(cap - 1, t_err.size());
    memcpy(buf, t_err.data(), k);
    buf[k] = 0;
  }
  return t_err.size();
}

int blsgpu_verify_batch(int sig_group, int scheme, const void* pks, const void* sigs, const uint8_t* msgs,
                        const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status) try {
  const bool wire = fmt == BLSGPU_FMT_COMPRESSED || fmt == BLSGPU_FMT_LEGACY;
  int rc = check_common(sig_group, scheme, wire ? BLSGPU_FMT_RAW_PROJ : fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!pks || !sigs || !msg_offsets || !status) return fail(BLSGPU_E_ARG, "null argument");
  if (const size_t D = shard_devices(n, {pks, sigs, msgs, msg_offsets, status}); D > 1) {
    // independent items: contiguous ranges per device, no exchange at all (message offsets are absolute, so every shard
    // passes the whole blob and its own window of the offsets)
    const size_t psz = pk_size(sig_group, fmt), ssz = sig_size(sig_group, fmt);
    return run_on_devices(D, [&](size_t d) {
      const size_t lo = n * d / D, hi = n * (d + 1) / D;
      return blsgpu_verify_batch(sig_group, scheme, (const uint8_t*)pks + lo * psz, (const uint8_t*)sigs + lo * ssz, msgs, msg_offsets + lo, hi - lo,
                                 fmt, status + lo);
    });
  }
  CTX_ACQUIRE(c);
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
  {
    const void* ptrs[4] = {pks, sigs, msgs, msg_offsets};
    const size_t sizes[4] = {pkb, sgb, (size_t)total, 8 * (n + 1)};
    const void* outs[4];
    if (stage_in_packed(c, ptrs, sizes, outs, 4)) {
      d_pks = outs[0];
      d_sigs = outs[1];
      d_msgs = outs[2];
      d_offs = outs[3];
    } else {
      if ((rc = stage_in(c, pks, pkb, &d_pks))) return rc;
      if ((rc = stage_in(c, sigs, sgb, &d_sigs))) return rc;
      if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
      if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
    }
  }
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
  if ((rc = status_out_and_sync(c, status, d_status, n))) return rc;
  return 0;
}
API_CATCH

/* Opt-in grouped verification of independent items (Bls12381G1Impl): groups of GROUPED_ITEMS items share one final
 * exponentiation through a random linear combination (csrc/kernels.cuh k_prepare_grouped); every group that fails -- and only
 * those -- is re-verified item by item with the kernels of blsgpu_verify_batch, so the status vector equals that call's except
 * that an invalid item whose group passes the combined check (probability below 2^-64 per group, over the seed) is reported
 * valid.  seed: the generator of the per-item scalars; the same seed gives the same run. */
int blsgpu_verify_batch_grouped(int sig_group, int scheme, const void* pks, const void* sigs, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n,
                                int fmt, uint64_t seed, int32_t* status) try {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (sig_group != 1) return fail(BLSGPU_E_ARG, "grouped verification is built for sig_group 1 (Bls12381G1Impl)");
  if (n == 0) return 0;
  if (!pks || !sigs || !msg_offsets || !status) return fail(BLSGPU_E_ARG, "null argument");
  if (n >= 0xffffffffull) return fail(BLSGPU_E_ARG, "more than 2^32 - 2 items");
  CTX_ACQUIRE(c);
  std::vector<uint64_t> offs_h(n + 1);
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(offs_h.data(), msg_offsets, 8 * (n + 1), hipMemcpyDeviceToHost));
  else memcpy(offs_h.data(), msg_offsets, 8 * (n + 1));
  const uint64_t total = offs_h[n];
  const size_t psz = pk_size(1, fmt), ssz = sig_size(1, fmt), pkb = psz * n, sgb = ssz * n;
  const size_t ng = (n + GROUPED_ITEMS - 1) / GROUPED_ITEMS, m = (GROUPED_ITEMS + 1) * ng;
  const size_t need = 2 * (pad256(pkb) + pad256(sgb) + pad256(total) + pad256(8 * (n + 1))) + 3 * pad256(4 * n) + pad256(144 * n) +
                      pad256((size_t)WS_PAIR1_WORDS * 4 * m) + pad256((size_t)WS_F_WORDS * 4 * m) + pad256(4 * m) + pad256((size_t)WS_F_WORDS * 4 * ng) +
                      pad256(4 * ng) + 2 * pad256((size_t)WS_PAIRS_WORDS * 4 * n) + 16384;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d_pks, *d_sigs, *d_msgs, *d_offs;
  if ((rc = stage_in(c, pks, pkb, &d_pks))) return rc;
  if ((rc = stage_in(c, sigs, sgb, &d_sigs))) return rc;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
  int32_t* d_status = (int32_t*)arena_take(c, 4 * n);
  uint8_t* d_scaled = (uint8_t*)arena_take(c, 144 * n);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIR1_WORDS * 4 * m);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * m);
  int32_t* d_skip = (int32_t*)arena_take(c, 4 * m);
  uint32_t* d_fg = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * ng);
  int32_t* d_gst = (int32_t*)arena_take(c, 4 * ng);
  if (!d_status || !d_scaled || !d_pairs || !d_f || !d_skip || !d_fg || !d_gst) return fail(BLSGPU_E_HIP, "internal: arena too small");
  const int aug = scheme == BLSGPU_SCHEME_AUG;
  const dst_arg dst = scheme_dst(1, scheme);
  HIPCK(hipMemsetAsync(d_skip, 0xff, 4 * m, c->stream));
  HIPCK(hipMemsetAsync(d_gst, 0, 4 * ng, c->stream));
  KL(KID_PREPARE, k_prepare_grouped<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_sigs, fmt, aug, (const uint8_t*)d_msgs,
     (const uint64_t*)d_offs, dst, (uint64_t)seed, ng, d_pairs, d_skip, d_scaled, d_status);
  KL(KID_ACCUM, k_group_sigsum, dim3(blocks_for(ng)), dim3(BLS_BLOCK), ng, n, (const uint8_t*)d_scaled, d_pairs, d_skip);
  MILLER1_LAUNCH(m, m, d_pairs, d_skip, d_f);
  KL(KID_F12_FOLD, k_f12_mul3, dim3(blocks_for(ng)), dim3(BLS_BLOCK), ng, (const uint32_t*)d_f, m, d_fg);
  KL(KID_FINALEXP_V1, k_finalexps, dim3(blocks_for(2 * ng)), dim3(BLS_BLOCK), ng, (const uint32_t*)d_fg, d_gst);
  HIPCK(hipGetLastError());
  std::vector<int32_t> gst(ng);
  HIPCK(hipMemcpyAsync(gst.data(), d_gst, 4 * ng, hipMemcpyDeviceToHost, c->stream));
  HIPCK(hipStreamSynchronize(c->stream));
  // the groups that failed: their items once more, one by one (identity failures keep the status they already have: the
  // per-item kernels find the same)
  std::vector<uint32_t> idx;
  for (size_t g = 0; g < ng; g++)
    if (gst[g] != BLS_OK)
      for (size_t i = g * GROUPED_ITEMS; i < n && i < (g + 1) * GROUPED_ITEMS; i++) idx.push_back((uint32_t)i);
  if (!idx.empty()) {
    const size_t cnt = idx.size();
    std::vector<uint64_t> offs2(cnt + 1);
    offs2[0] = 0;
    for (size_t j = 0; j < cnt; j++) offs2[j + 1] = offs2[j] + (offs_h[idx[j] + 1] - offs_h[idx[j]]);
    uint32_t* d_idx = (uint32_t*)arena_take(c, 4 * cnt);
    uint64_t* d_offs2 = (uint64_t*)arena_take(c, 8 * (cnt + 1));
    uint8_t* d_pks2 = (uint8_t*)arena_take(c, psz * cnt);
    uint8_t* d_sigs2 = (uint8_t*)arena_take(c, ssz * cnt);
    uint8_t* d_msgs2 = (uint8_t*)arena_take(c, offs2[cnt] ? offs2[cnt] : 1);
    int32_t* d_st2 = (int32_t*)arena_take(c, 4 * cnt);
    uint32_t* d_pairs2 = (uint32_t*)arena_take(c, (size_t)WS_PAIRS_WORDS * 4 * cnt);
    uint32_t* d_f2 = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * cnt);
    if (!d_idx || !d_offs2 || !d_pks2 || !d_sigs2 || !d_msgs2 || !d_st2 || !d_pairs2 || !d_f2) return fail(BLSGPU_E_HIP, "internal: arena too small");
    HIPCK(hipMemcpyAsync(d_idx, idx.data(), 4 * cnt, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(d_offs2, offs2.data(), 8 * (cnt + 1), hipMemcpyHostToDevice, c->stream));
    KL(KID_COMPRESS, k_gather_rows, dim3(blocks_for(cnt * (psz / 4))), dim3(BLS_BLOCK), cnt, (const uint32_t*)d_idx, (const uint32_t*)d_pks, psz / 4, (uint32_t*)d_pks2);
    KL(KID_COMPRESS, k_gather_rows, dim3(blocks_for(cnt * (ssz / 4))), dim3(BLS_BLOCK), cnt, (const uint32_t*)d_idx, (const uint32_t*)d_sigs, ssz / 4, (uint32_t*)d_sigs2);
    KL(KID_COMPRESS, k_gather_ragged, dim3(blocks_for(cnt)), dim3(BLS_BLOCK), cnt, (const uint32_t*)d_idx, (const uint64_t*)d_offs, (const uint64_t*)d_offs2,
       (const uint8_t*)d_msgs, d_msgs2);
    HIPCK(hipGetLastError());
    if ((rc = run_verify_items(c, 1, aug, d_pks2, d_sigs2, fmt, d_msgs2, d_offs2, 0, dst, cnt, d_pairs2, d_f2, d_st2))) {
      (void)hipStreamSynchronize(c->stream);      // idx / offs2 are read by the copies above
      return rc;
    }
    KL(KID_COMPRESS, k_scatter_i32, dim3(blocks_for(cnt)), dim3(BLS_BLOCK), cnt, (const uint32_t*)d_idx, (const int32_t*)d_st2, d_status);
    HIPCK(hipGetLastError());
  }
  if ((rc = status_out_and_sync(c, status, d_status, n))) return rc;
  return 0;
}
API_CATCH

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
  const uint64_t offs_h[2] = {0, (uint64_t)msg_len};
  uint64_t* d_offs = (uint64_t*)arena_take(c, 16);
  int32_t* d_status = (int32_t*)arena_take(c, 4);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, WS_PAIRS_WORDS * 4);
  uint32_t* d_f = (uint32_t*)arena_take(c, WS_F_WORDS * 4);
  uint8_t* d_sig_proj = (uint8_t*)arena_take(c, 288);
  if (!d_offs || !d_status || !d_pairs || !d_f || !d_sig_proj) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if ((rc = h2d_small(c, d_offs, offs_h, 16))) return rc;
  // normalise the signature to RAW_PROJ with a 1-point "sum"
  if (sig_group == 1) rc = run_point_sum<1>(c, (const uint8_t*)d_sig_in, fmt, nullptr, nullptr, 1, d_sig_proj, 1);
  else rc = run_point_sum<2>(c, (const uint8_t*)d_sig_in, fmt, nullptr, nullptr, 1, d_sig_proj, 1);
  if (rc) return rc;
  if (d_hash) {
    if (sig_group == 1) KL(KID_PREPARE, k_prepare_hashed<1>, dim3(1), dim3(BLS_BLOCK), (size_t)1, d_pk_proj, (const uint8_t*)d_sig_proj, d_hash, d_pairs, d_status, 0);
    else KL(KID_PREPARE, k_prepare_hashed<2>, dim3(1), dim3(BLS_BLOCK), (size_t)1, d_pk_proj, (const uint8_t*)d_sig_proj, d_hash, d_pairs, d_status, 0);
    rc = run_pairing2(c, 1, d_pairs, d_f, d_status, sig_group == 1 ? 1 : 0);
  } else {
    rc = run_verify_items(c, sig_group, aug_prefix, d_pk_proj, d_sig_proj, BLSGPU_FMT_RAW_PROJ, (const uint8_t*)d_msg, d_offs, 1,
                          scheme_dst(sig_group, scheme), 1, d_pairs, d_f, d_status);
  }
  if (rc) return rc;
  if ((rc = status_out_and_sync(c, status, d_status, 1))) return rc;
  return 0;
}

// The one core_verify at the end of MultiSignature::verify / verify_secure for Bls12381G1Impl when the message takes no key
// prefix, cut in time (k_pairing_pre / k_pairing_post): BEGIN runs on the side stream, beside the whole key sum -- the hash of
// the message, the signature's identity check, the Miller function of the (signature, -g2) pair; FINISH runs once the summed
// key exists: its identity check and affine form, its line coefficients, the other Miller function, the final exponentiation.
struct CutTail {
  uint32_t* rec = nullptr;
  int32_t* d_status = nullptr;
  int sg = 1;
};
static int cut_tail_begin(Ctx* c, int sig_group, int scheme, const void* sig, int fmt, const uint8_t* msg, size_t msg_len, CutTail& t) {
  int rc;
  const void *d_sig, *d_msg0;
  t.sg = sig_group;
  if ((rc = stage_in(c, sig, sig_size(sig_group, fmt), &d_sig))) return rc;
  if ((rc = stage_in(c, msg, msg_len, &d_msg0))) return rc;
  uint64_t* d_offs0 = (uint64_t*)arena_take(c, 16);
  t.rec = (uint32_t*)arena_take(c, (size_t)WREC_WORDS * 4);
  t.d_status = (int32_t*)arena_take(c, 4);
  uint8_t* d_hash = (uint8_t*)arena_take(c, 288);
  if (!d_offs0 || !t.rec || !t.d_status || !d_hash) return fail(BLSGPU_E_HIP, "internal: arena too small");
  const uint64_t offs0[2] = {0, (uint64_t)msg_len};
  if ((rc = h2d_small(c, d_offs0, offs0, 16))) return rc;
  if ((rc = side_fork(c))) return rc;
  if (sig_group == 1) {
    launch_hash_g1_small(c->side, 1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, 1, scheme_dst(1, scheme), (uint8_t*)nullptr, t.rec);
    hipLaunchKernelGGL(k_prepare_keys<1>, dim3(1), dim3(BLS_BLOCK), 0, c->side2, (size_t)1, (const uint8_t*)nullptr, (const uint8_t*)d_sig,
                       (const uint8_t*)nullptr, fmt, 1, t.rec, t.d_status);
    hipLaunchKernelGGL(k_pairing_pre, dim3(1, 1), dim3(WIDE_ENGINE_BLOCK), 0, c->side2, (size_t)1, t.rec, (const int32_t*)t.d_status, 1, 1);
  } else {
    // Bls12381G2Impl: both G2 points are known now (H(m) after its hash, the signature at once): the side stream hashes and
    // derives the lines of H(m); the second one runs the signature's lines and the Miller function of (-g1, signature).  The
    // summed key is a G1 point and only scales lines: it enters in POST.
    hipLaunchKernelGGL(k_prepare_keys<2>, dim3(1), dim3(BLS_BLOCK), 0, c->side2, (size_t)1, (const uint8_t*)nullptr, (const uint8_t*)d_sig,
                       (const uint8_t*)nullptr, fmt, 1, t.rec, t.d_status);
    hipLaunchKernelGGL(k_pairing_pre, dim3(1, 1), dim3(WIDE_ENGINE_BLOCK), 0, c->side2, (size_t)1, t.rec, (const int32_t*)t.d_status, 2, 2);
    const hipError_t ea = hipEventRecord(c->ev_join2, c->side2);
    if (!launch_hash_g2_small(c->side, 1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, 1, scheme_dst(2, scheme), d_hash, nullptr, t.rec))
      hipLaunchKernelGGL(k_prepare_keys<2>, dim3(1), dim3(BLS_BLOCK), 0, c->side, (size_t)1, (const uint8_t*)nullptr, (const uint8_t*)nullptr,
                         (const uint8_t*)d_hash, 0, 4, t.rec, t.d_status);
    const hipError_t eb = hipStreamWaitEvent(c->side, c->ev_join2, 0);      // the lines of H(m) read the status the signature part set
    hipLaunchKernelGGL(k_pairing_pre, dim3(1, 1), dim3(WIDE_ENGINE_BLOCK), 0, c->side, (size_t)1, t.rec, (const int32_t*)t.d_status, 0, 0);
    if (ea != hipSuccess || eb != hipSuccess) {
      (void)hipStreamSynchronize(c->side);
      (void)hipStreamSynchronize(c->side2);
      return fail(BLSGPU_E_HIP, "side-stream launch failed");
    }
  }
  const hipError_t e1 = hipGetLastError(), e2 = hipEventRecord(c->ev_join, c->side), e3 = hipEventRecord(c->ev_join2, c->side2);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    (void)hipStreamSynchronize(c->side);
    (void)hipStreamSynchronize(c->side2);
    return fail(BLSGPU_E_HIP, "side-stream launch failed");
  }
  return 0;
}
// rc: the caller's error so far (the side streams are joined either way: their kernels read the arena)
static int cut_tail_finish(Ctx* c, int rc, const uint8_t* d_pk_proj, CutTail& t, int32_t* status) {
  const hipError_t e = hipStreamWaitEvent(c->stream, c->ev_join, 0), e2 = hipStreamWaitEvent(c->stream, c->ev_join2, 0);
  if (rc || e != hipSuccess || e2 != hipSuccess) {
    (void)hipStreamSynchronize(c->side);
    (void)hipStreamSynchronize(c->side2);
    return rc ? rc : fail(BLSGPU_E_HIP, "hipStreamWaitEvent failed");
  }
  if (t.sg == 1) {
    KL(KID_PREPARE, k_prepare_keys<1>, dim3(1), dim3(BLS_BLOCK), (size_t)1, d_pk_proj, (const uint8_t*)nullptr, (const uint8_t*)nullptr, BLSGPU_FMT_RAW_PROJ, 2,
       t.rec, t.d_status);
    // the summed key's lines beside the Miller loop that consumes them
    if ((rc = launch_lines_and_post(c, 1, t.rec, t.d_status))) {
      (void)hipStreamSynchronize(c->side);
      (void)hipStreamSynchronize(c->side2);
      return rc;
    }
  } else {
    KL(KID_PREPARE, k_prepare_keys<2>, dim3(1), dim3(BLS_BLOCK), (size_t)1, d_pk_proj, (const uint8_t*)nullptr, (const uint8_t*)nullptr, BLSGPU_FMT_RAW_PROJ, 2,
       t.rec, t.d_status);
    if ((rc = launch_post(c, 1, t.rec, t.d_status))) {
      (void)hipStreamSynchronize(c->side);
      (void)hipStreamSynchronize(c->side2);
      return rc;
    }
  }
  HIPCK(hipGetLastError());
  if ((rc = status_out_and_sync(c, status, t.d_status, 1))) return rc;
  return 0;
}
static bool cut_tail_applies(int sig_group, int scheme) {
  (void)sig_group;
  return scheme != BLSGPU_SCHEME_AUG && wide_max_items() >= 1 && coop_max_items() >= 1;
}

int blsgpu_multi_verify(int sig_group, int scheme, const void* pks, size_t n, const void* sig, const uint8_t* msg,
                        size_t msg_len, int fmt, int32_t* status) try {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!sig || !status || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  if (const size_t D = shard_devices(n, {pks, sig}); D > 1) {
    // per-device partial key sums (reference src/traits/pk_multi.rs:7-13 cut into ranges), then the fold and the one
    // verification on device 0: MultiSignature::verify over the D partial sums
    const size_t psz = pk_size(sig_group, fmt), osz = sig_group == 1 ? 288 : 144;
    std::vector<uint8_t> parts(osz * D), sig_proj(288);
    if ((rc = run_on_devices(D, [&](size_t d) {
          const size_t lo = n * d / D, hi = n * (d + 1) / D;
          return (sig_group == 1 ? blsgpu_sum_g2 : blsgpu_sum_g1)((const uint8_t*)pks + lo * psz, hi - lo, fmt, parts.data() + osz * d);
        })))
      return rc;
    NestedScope ns;
    if ((rc = (sig_group == 1 ? blsgpu_sum_g1 : blsgpu_sum_g2)(sig, 1, fmt, sig_proj.data()))) return rc;   // the signature as RAW_PROJ
    return blsgpu_multi_verify(sig_group, scheme, parts.data(), D, sig_proj.data(), msg, msg_len, BLSGPU_FMT_RAW_PROJ, status);
  }
  CTX_ACQUIRE(c);
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
  CutTail cut;
  const bool use_cut = cut_tail_applies(sig_group, scheme) && n > 0;
  if (use_cut) {
    if ((rc = cut_tail_begin(c, sig_group, scheme, sig, fmt, msg, msg_len, cut))) return rc;
  } else if (scheme != BLSGPU_SCHEME_AUG && n > 0) {
    const void* d_msg0;
    if ((rc = stage_in(c, msg, msg_len, &d_msg0))) return rc;
    uint64_t* d_offs0 = (uint64_t*)arena_take(c, 16);
    d_hash = (uint8_t*)arena_take(c, 288);
    if (!d_offs0 || !d_hash) return fail(BLSGPU_E_HIP, "internal: arena too small");
    const uint64_t offs0[2] = {0, (uint64_t)msg_len};
    if ((rc = h2d_small(c, d_offs0, offs0, 16))) return rc;
    if ((rc = side_fork(c))) return rc;
    if (sig_group == 1)   // one wave, the two SSWU maps on two DPP rows in the row-wide field type (csrc/wide.cuh)
      launch_hash_g1_small(c->side, 1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, 1, scheme_dst(sig_group, scheme), d_hash, (uint32_t*)nullptr);
    else
      launch_hash_g2_small(c->side, 1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, 1, scheme_dst(sig_group, scheme), d_hash);
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(c->ev_join, c->side));
  }
  // MultiPublicKey::from_public_keys: the serial `g += key` of reference src/traits/pk_multi.rs:7-13 as a tree sum
  if (sig_group == 1) rc = run_point_sum<2>(c, (const uint8_t*)d_pks, fmt, nullptr, nullptr, n, d_part, T);
  else rc = run_point_sum<1>(c, (const uint8_t*)d_pks, fmt, nullptr, nullptr, n, d_part, T);
  if (use_cut) return cut_tail_finish(c, rc, d_part, cut, status);
  if (d_hash) {
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_join, 0);    // also on the error path: the side kernel reads the arena
    if (rc) (void)hipStreamSynchronize(c->side);
    if (e != hipSuccess) return fail(BLSGPU_E_HIP, "hipStreamWaitEvent failed");
  }
  if (rc) return rc;
  return verify_one_tail(c, sig_group, scheme, scheme == BLSGPU_SCHEME_AUG, d_part, sig, fmt, msg, msg_len, status, d_hash);
}
API_CATCH

// The device part of core_aggregate_verify (reference src/traits/sig_core.rs:149-178) for n (pk, msg) pairs [plus the
// (sig, -g) pair when d_sig != nullptr], enqueued without a host round trip: hash-to-curve + identity flags, the first
// identity key reduced on the device (d_first: index, n when the signature is the identity, -1 when none), Miller loops
// (flagged pairs contribute 1), product tree; then either the final exponentiation's verdict (d_verdict) or the product
// before the final exponentiation as a 576-byte record (d_rec).
static int aggregate_enqueue(Ctx* c, int sig_group, int scheme, const uint8_t* d_pks, const uint8_t* d_sig, int fmt, const uint8_t* d_msgs,
                             const uint64_t* d_offs, size_t n, int64_t* d_first, int32_t* d_verdict, uint8_t* d_rec) {
  int rc;
  const size_t m = n + 1, mm = d_sig ? m : n;
  const int has_sig = d_sig ? 1 : 0;
  int32_t* d_bad = (int32_t*)arena_take(c, 4 * m);
  unsigned long long* d_min = (unsigned long long*)arena_take(c, 64);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIR1_WORDS * 4 * m);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * m);
  if (!d_bad || !d_min || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
  HIPCK(hipMemsetAsync(d_min, 0xff, 8, c->stream));
  HIPCK(hipMemsetAsync(d_bad, 0, 4 * m, c->stream));
  const dst_arg dst = scheme_dst(sig_group, scheme);
  const int aug = scheme == BLSGPU_SCHEME_AUG;
  // Bls12381G1Impl: the messages are hashed to E1(Fp) WITHOUT the cofactor clearing (a third of the hash) and the signature's
  // pair is (sig, -[c] g2) instead of (sig, -g2): the same verdict (csrc/g2neg_lines.cuh), also shard by shard -- only the
  // product of all records is ever exponentiated
  // ... on two lanes per item (the two SSWU maps of the hash side by side) while that fits one machine round of lanes: the kernel is a
  // latency chain there (1.75 -> 1.1 ms at 32,768 pairs, 1.63 -> 0.96 at 4,096); beyond a round the redundant rest of the two-lane form costs more
  // than the shorter chains save (5.9 against 5.3 ms at 262,144).  BLSGPU_AGG_LANES=1 / 2 forces a form (A/B)
  const int agg_lanes_env = (int)knobs().agg_lanes;
  const int agg_one_lane = agg_lanes_env == 1 || (agg_lanes_env != 2 && mm > 65536 + 1024);
  const int agg_flags = sig_group == 1 ? (agg_one_lane ? 2 : 3) : 1;
  if (mm > 0) {
    // the workspace stride is always n + 1 (k_prepare_agg's layout); without a signature lane n stays idle
    if (sig_group == 1)
      KL(KID_PREPARE_AGG, k_prepare_agg<1>, dim3(blocks_for(agg_one_lane ? mm : 2 * mm)), dim3(BLS_BLOCK), n, d_pks, d_sig, fmt, aug, d_msgs, d_offs, dst, d_pairs, d_bad, agg_flags, has_sig);
    else
      KL(KID_PREPARE_AGG, k_prepare_agg<2>, dim3(blocks_for(2 * mm)), dim3(BLS_BLOCK), n, d_pks, d_sig, fmt, aug, d_msgs, d_offs, dst, d_pairs, d_bad, agg_flags, has_sig);
    if (n) KL(KID_FIRST_BAD, k_first_bad, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const int32_t*)d_bad, d_min);
  }
  KL(KID_FIRST_BAD, k_first_bad_fin, dim3(1), dim3(BLS_BLOCK), n, (const int32_t*)d_bad, has_sig, (const unsigned long long*)d_min, d_first);
  HIPCK(hipGetLastError());
  if (mm == 0) {                   // an empty shard: the neutral record
    if (d_rec) {
      uint8_t one[576];
      memset(one, 0, sizeof one);
      memcpy(one, FP_ONE_HOST, 48);
      if ((rc = h2d_small(c, d_rec, one, 576))) return rc;
    }
    if (d_verdict) HIPCK(hipMemsetAsync(d_verdict, 0, 4, c->stream));   // the empty product is one
    return 0;
  }
  size_t outputs = 0;
  // the signature's pair (item n, when present): its G2 member is the constant -[c] g2 for Bls12381G1Impl, the signature itself otherwise
  if ((rc = run_miller_product(c, mm, m, d_pairs, d_bad, d_f, &outputs, has_sig ? (sig_group == 1 ? ((agg_flags & 2) ? 2 : 1) : 3) : 0))) return rc;
  if (d_verdict) {
    if ((rc = run_f12_product_verdict(c, d_f, outputs, m, d_verdict))) return rc;
  } else {
    if ((rc = run_f12_fold(c, d_f, outputs, m))) return rc;
    KL(KID_F12_IO, k_f12_export, dim3(1), dim3(BLS_BLOCK), d_f, m, d_rec);
    HIPCK(hipGetLastError());
  }
  return 0;
}

// reference src/traits/sig_basic.rs:46-58 on host-resident messages: first i whose message equals an earlier one.
// Open-addressing table of message indices keyed by a 64-bit hash (computed on all host cores), equality verified on the
// bytes; the scan is sequential so that the FIRST such i is reported, as the reference does.  Returns true on a duplicate.
static bool duplicate_check_host(const uint8_t* mh, const uint64_t* offs_h, size_t n, uint64_t aux_h[2]) {
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
      std::vector<JoinedThread> pool;
      pool.reserve(nthr);
      for (unsigned t = 0; t < nthr; t++) pool.emplace_back([&work, t] { work(t); });
      for (auto& th : pool) th.join();
    }
  }
  size_t cap = 16;
  while (cap < 2 * n) cap <<= 1;
  std::vector<uint32_t> tab(cap, 0);          // index + 1, 0 = empty
  for (size_t i = 0; i < n; i++) {
    const size_t len = (size_t)(offs_h[i + 1] - offs_h[i]);
    size_t slot = (size_t)hs[i] & (cap - 1);
    while (tab[slot]) {
      const size_t j = tab[slot] - 1;
      if (hs[j] == hs[i] && (size_t)(offs_h[j + 1] - offs_h[j]) == len && memcmp(mh + offs_h[j], mh + offs_h[i], len) == 0) {
        aux_h[0] = j;
        aux_h[1] = i;
        return true;
      }
      slot = (slot + 1) & (cap - 1);
    }
    tab[slot] = (uint32_t)(i + 1);
  }
  return false;
}

int blsgpu_aggregate_verify(int sig_group, int scheme, const void* pks, const uint8_t* msgs, const uint64_t* msg_offsets,
                            size_t n, const void* sig, int fmt, int32_t* status, uint64_t* aux) try {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!sig || !status || !msg_offsets || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  if (const size_t D = shard_devices(n, {pks, msgs, msg_offsets, sig}); D > 1) {
    // every device: hash-to-curve + Miller loops of its range -> one Fp12 record (blsgpu_aggregate_partial; device 0 also
    // takes the signature pair); meanwhile this thread applies Basic's duplicate rule to the whole list; then the D records
    // are multiplied and exponentiated once on device 0 -- the exchange that the one-process-per-GPU path does over RCCL
    const size_t psz = pk_size(sig_group, fmt);
    std::vector<uint8_t> recs(576 * D);
    std::vector<int64_t> fbs(D, -1);
    uint64_t dup[2] = {~0ull, ~0ull};
    int dup_rc = 0;
    std::string dup_err;
    JoinedThread dup_thread;
    if (scheme == BLSGPU_SCHEME_BASIC)
      dup_thread = JoinedThread([&] {
        t_nested = true;
        if (is_device_ptr(msgs)) {
          dup_rc = blsgpu_first_duplicate_message(msgs, msg_offsets, n, dup);
        } else {
          std::vector<uint64_t> oh;
          const uint64_t* offs = msg_offsets;
          if (is_device_ptr(msg_offsets)) {
            oh.resize(n + 1);
            if (hipMemcpy(oh.data(), msg_offsets, 8 * (n + 1), hipMemcpyDeviceToHost) != hipSuccess) dup_rc = fail(BLSGPU_E_HIP, "copying the message offsets");
            offs = oh.data();
          }
          uint64_t a2[2] = {0, 0};
          if (!dup_rc && duplicate_check_host(msgs, offs, n, a2)) {
            dup[0] = a2[0];
            dup[1] = a2[1];
          }
        }
        dup_err = t_err;
      });
    rc = run_on_devices(D, [&](size_t d) {
      const size_t lo = n * d / D, hi = n * (d + 1) / D;
      return blsgpu_aggregate_partial(sig_group, scheme, (const uint8_t*)pks + lo * psz, msgs, msg_offsets + lo, hi - lo, d == 0 ? sig : nullptr, fmt,
                                      recs.data() + 576 * d, &fbs[d]);
    });
    dup_thread.join();
    if (rc) return rc;
    if (dup_rc) {
      t_err = dup_err;
      return dup_rc;
    }
    int32_t st = BLSGPU_OK;
    uint64_t aux_h[2] = {0, 0};
    if (dup[1] != ~0ull) {
      st = BLSGPU_DUPLICATE_MESSAGE;
      aux_h[0] = dup[0];
      aux_h[1] = dup[1];
    } else if (fbs[0] == (int64_t)(n * 1 / D)) {          // device 0 saw the identity signature (its local index n)
      st = BLSGPU_SIG_IDENTITY;
    } else {
      for (size_t d = 0; d < D && st == BLSGPU_OK; d++)
        if (fbs[d] >= 0) {
          st = BLSGPU_PK_IDENTITY;
          aux_h[0] = n * d / D + (uint64_t)fbs[d] + 1;
        }
      if (st == BLSGPU_OK) {
        NestedScope ns;
        int32_t one = 0;
        if ((rc = blsgpu_fp12_product_is_one(recs.data(), D, &one))) return rc;
        st = one ? BLSGPU_OK : BLSGPU_INVALID_SIGNATURE;
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
  CTX_ACQUIRE(c);
  uint64_t aux_h[2] = {0, 0};
  const bool trace = knobs().host_trace != 0;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  const bool basic = scheme == BLSGPU_SCHEME_BASIC;
  // Basic's duplicate-message rule (reference src/traits/sig_basic.rs:46-58) runs where the messages live: on the device for
  // device-resident messages (k_dup_*), on the host (overlapped with the device's hashing) for host buffers.  In the
  // reference it runs to completion before anything else; here the verdict keeps that precedence.
  const bool dev_msgs = is_device_ptr(msgs), dev_offs = is_device_ptr(msg_offsets);
  const bool dup_on_device = basic && dev_msgs;
  std::vector<uint64_t> offs_h;
  uint64_t total = 0;
  if (basic && !dev_msgs && dev_offs) {   // host messages with device offsets: the host check needs the offsets
    offs_h.resize(n + 1);
    HIPCK(hipMemcpy(offs_h.data(), msg_offsets, 8 * (n + 1), hipMemcpyDeviceToHost));
    total = offs_h[n];
  } else if (dev_offs) {
    HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  } else {
    total = msg_offsets[n];
  }
  const size_t m = n + 1, psz = pk_size(sig_group, fmt);
  size_t need = pad256(psz * n) + pad256(sig_size(sig_group, fmt)) + pad256(total) + pad256(8 * m) + pad256(4 * m) +
                2 * pad256((size_t)WS_PAIRS_WORDS * 4 * m) + 8192 + (dup_on_device ? dup_ws_bytes(n) : 0);
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d_pks, *d_sig, *d_msgs, *d_offs;
  if ((rc = stage_in(c, pks, psz * n, &d_pks))) return rc;
  if ((rc = stage_in(c, sig, sig_size(sig_group, fmt), &d_sig))) return rc;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * m, &d_offs))) return rc;
  const double t_staged = now();
  int64_t* d_first = (int64_t*)arena_take(c, 64);
  int32_t* d_verdict = (int32_t*)arena_take(c, 64);
  uint64_t* d_dup = (uint64_t*)arena_take(c, 64);
  struct results { int64_t first; int32_t verdict; int32_t pad; uint64_t dup[2]; };
  results* h = (results*)hsmall_take(c, sizeof(results));
  if (!d_first || !d_verdict || !d_dup || !h) return fail(BLSGPU_E_HIP, "internal: arena too small");
  h->dup[0] = h->dup[1] = ~0ull;
  // everything is enqueued at once; the identity index, the verdict and the duplicate pair come back together and the
  // host applies the reference's precedence afterwards
  if ((rc = aggregate_enqueue(c, sig_group, scheme, (const uint8_t*)d_pks, (const uint8_t*)d_sig, fmt, (const uint8_t*)d_msgs,
                              (const uint64_t*)d_offs, n, d_first, d_verdict, nullptr)))
    return rc;
  if (dup_on_device) {
    if ((rc = run_first_duplicate(c, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, n, d_dup))) return rc;
    HIPCK(hipMemcpyAsync(h->dup, d_dup, 16, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCK(hipMemcpyAsync(&h->first, d_first, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCK(hipMemcpyAsync(&h->verdict, d_verdict, 4, hipMemcpyDeviceToHost, c->stream));
  bool dup = false;
  if (basic && !dev_msgs) {       // on the host, while the device works
    dup = duplicate_check_host(msgs, offs_h.empty() ? msg_offsets : offs_h.data(), n, aux_h);
    if (trace) fprintf(stderr, "[blsgpu] aggregate_verify n=%zu: host duplicate check %.2f ms (overlapped)\n", n, now() - t_staged);
  }
  SYNC_FLUSH(c);
  if (trace) fprintf(stderr, "[blsgpu] aggregate_verify n=%zu: staging %.2f ms, total %.2f ms\n", n, t_staged - t_start, now() - t_start);
  if (dup_on_device && h->dup[1] != ~0ull) {
    dup = true;
    aux_h[0] = h->dup[0];
    aux_h[1] = h->dup[1];
  }
  int32_t st;
  if (dup) {
    st = BLSGPU_DUPLICATE_MESSAGE;   // reported before any identity check (sig_basic.rs:46-58 precedes core_aggregate_verify)
  } else if (h->first == (int64_t)n) {
    st = BLSGPU_SIG_IDENTITY;        // reference src/traits/sig_core.rs:155-167: signature identity first, ...
  } else if (h->first >= 0) {
    st = BLSGPU_PK_IDENTITY;         // ... then the first identity key (1-based)
    aux_h[0] = (uint64_t)h->first + 1;
  } else {
    st = h->verdict;
  }
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  if (aux) {
    if (is_device_ptr(aux)) HIPCK(hipMemcpy(aux, aux_h, 16, hipMemcpyHostToDevice));
    else memcpy(aux, aux_h, 16);
  }
  return 0;
}
API_CATCH

// sig is the identity <=> Ok for an empty key list (reference src/secure_aggregation.rs:189-195)
static int empty_list_verdict(Ctx* c, int sig_group, const void* sig, int fmt, int32_t* st) {
  int rc;
  const void* d_sig;
  if ((rc = stage_in(c, sig, sig_size(sig_group, fmt), &d_sig))) return rc;
  uint8_t* d_proj = (uint8_t*)arena_take(c, 288);
  uint8_t* h = (uint8_t*)hsmall_take(c, 288);
  if (!d_proj || !h) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if (sig_group == 1) rc = run_point_sum<1>(c, (const uint8_t*)d_sig, fmt, nullptr, nullptr, 1, d_proj, 1);
  else rc = run_point_sum<2>(c, (const uint8_t*)d_sig, fmt, nullptr, nullptr, 1, d_proj, 1);
  if (rc) return rc;
  HIPCK(hipMemcpyAsync(h, d_proj, 288, hipMemcpyDeviceToHost, c->stream));
  SYNC_FLUSH(c);
  const size_t zoff = sig_group == 1 ? 96 : 192, zlen = sig_group == 1 ? 48 : 96;
  bool inf = true;
  for (size_t k = 0; k < zlen; k++) inf = inf && h[zoff + k] == 0;
  *st = inf ? BLSGPU_OK : BLSGPU_INVALID_SIGNATURE;
  return 0;
}

int blsgpu_verify_secure(int sig_group, int scheme, const void* pks, size_t n, const void* sig, const uint8_t* msg,
                         size_t msg_len, int ser_format, int fmt, int32_t* status) try {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!sig || !status || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  if (ser_format != 0 && ser_format != 1) return fail(BLSGPU_E_ARG, "ser_format must be 0 (Modern) or 1 (Legacy)");
  if (ser_format == 1 && sig_group != 2)
    return fail(BLSGPU_E_ARG, "Legacy serialization exists only for Bls12381G2Impl (48-byte keys), reference src/signature.rs:201-204");
  if (n >= 0xffffffffull) return fail(BLSGPU_E_ARG, "more than 2^32 - 2 keys");
  if (const size_t D = shard_devices(n, {pks, sig}); D > 1) {
    // every device serialises its range of the keys; device 0 sorts all of them and a host core hashes the sorted stream
    // (the one sequential step, reference src/secure_aggregation.rs:45-59); every device derives the coefficients of its
    // own keys and adds up its share of sum t_i pk_i; device 0 folds the D partial sums and runs the one verification
    const int pk_group = sig_group == 1 ? 2 : 1;
    const size_t psz = pk_size(sig_group, fmt), width = sig_group == 1 ? 96 : 48, osz = sig_group == 1 ? 288 : 144;
    std::vector<uint8_t> kb(width * n), parts(osz * D), apk(288), sig_proj(288), hm(288), digest(32);
    std::vector<uint32_t> perm(n);
    std::vector<int32_t> sts(D, 0);
    const char* dsts = DST_TABLE[sig_group - 1][scheme];
    const uint64_t offs1[2] = {0, (uint64_t)msg_len};
    int hash_rc = 0;
    std::string hash_err;
    JoinedThread hash_thread([&] {           // H(msg) beside everything else (a second context of device 0)
      t_nested = true;
      hash_rc = (sig_group == 1 ? blsgpu_hash_to_g1 : blsgpu_hash_to_g2)(msg, offs1, 1, (const uint8_t*)dsts, strlen(dsts), hm.data());
      hash_err = t_err;
    });
    rc = run_on_devices(D, [&](size_t d) {
      const size_t lo = n * d / D, hi = n * (d + 1) / D;
      return blsgpu_serialize(pk_group, (const uint8_t*)pks + lo * psz, hi - lo, fmt, ser_format ? BLSGPU_FMT_LEGACY : BLSGPU_FMT_COMPRESSED,
                              kb.data() + width * lo, nullptr);
    });
    if (!rc) {
      NestedScope ns;
      rc = blsgpu_sort_keys(kb.data(), n, width, perm.data());
      if (!rc) rc = blsgpu_sorted_keys_digest(kb.data(), perm.data(), n, width, digest.data());
    }
    if (!rc)
      rc = run_on_devices(D, [&](size_t d) {
        const size_t lo = n * d / D, hi = n * (d + 1) / D;
        std::vector<uint8_t> scal(32 * (hi - lo));
        int r2 = blsgpu_coefficients_for_range(digest.data(), perm.data(), n, lo, hi - lo, scal.data(), &sts[d]);
        if (r2) return r2;
        return (pk_group == 1 ? blsgpu_msm_g1 : blsgpu_msm_g2)((const uint8_t*)pks + lo * psz, scal.data(), hi - lo, fmt, parts.data() + osz * d);
      });
    hash_thread.join();
    if (rc) return rc;
    if (hash_rc) {
      t_err = hash_err;
      return hash_rc;
    }
    int32_t st = BLSGPU_OK;
    for (size_t d = 0; d < D; d++)
      if (sts[d] != BLSGPU_OK) st = BLSGPU_INVALID_COEFFICIENT;
    if (st == BLSGPU_OK) {
      NestedScope ns;
      if ((rc = (pk_group == 1 ? blsgpu_sum_g1 : blsgpu_sum_g2)(parts.data(), D, BLSGPU_FMT_RAW_PROJ, apk.data()))) return rc;
      if ((rc = (sig_group == 1 ? blsgpu_sum_g1 : blsgpu_sum_g2)(sig, 1, fmt, sig_proj.data()))) return rc;
      if ((rc = blsgpu_core_verify_hashed(sig_group, apk.data(), sig_proj.data(), hm.data(), 1, &st))) return rc;
    }
    if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
    else *status = st;
    return 0;
  }
  CTX_ACQUIRE(c);
  const bool trace = knobs().host_trace != 0;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const size_t psz = pk_size(sig_group, fmt), width = sig_group == 1 ? 96 : 48, T = accumulate_lanes(n);
  // weighted MSM tables (csrc/kernels.cuh k_msm2_tables): built while the host hashes the key stream; BLSGPU_MSM2_TABLES=0 switches them off
  // (up to 262,144 keys: the tables are 4.5 KB per G2 key -- 1.2 GB there -- and stay in the context's arena; beyond that size the sum
  // is throughput-bound and its chunk lanes' latency no longer shows)
  const bool msm_tables = knobs().msm2_tables != 0 && msm_use_pippenger(n) && !msm_use_v1() && n <= ((size_t)1 << 18);
  size_t need = pad256(psz * n) + pad256(width * n) + pad256(32 * n) + pad256(288 * T) + pad256(msg_len) + 16384 + msm_ws_bytes(n) +
                keysort_ws_bytes(n, width) + (msm_tables ? msm2_tables_bytes(n, sig_group == 1 ? 2 : 1) : 0);
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  int32_t st = BLSGPU_OK;
  if (n == 0) {
    if ((rc = empty_list_verdict(c, sig_group, sig, fmt, &st))) return rc;
    if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
    else *status = st;
    return 0;
  }
  const void* d_pks;
  if ((rc = stage_in(c, pks, psz * n, &d_pks))) return rc;
  uint8_t* d_bytes = (uint8_t*)arena_take(c, width * n);
  uint8_t* d_scal = (uint8_t*)arena_take(c, 32 * n);
  uint8_t* d_part = (uint8_t*)arena_take(c, 288 * T);
  uint8_t* d_H = (uint8_t*)arena_take(c, 64);
  int32_t* d_zero = (int32_t*)arena_take(c, 64);
  int32_t* h_zero = (int32_t*)hsmall_take(c, 64);
  keysort_ws w;
  if (!d_bytes || !d_scal || !d_part || !d_H || !d_zero || !h_zero) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if ((rc = keysort_ws_take(c, n, width, w))) return rc;
  const double t0 = now();
  // PublicKey::to_bytes / to_bytes_with_mode of every key (reference src/secure_aggregation.rs:42,47; public_key.rs:146-151)
  if (sig_group == 1)
    KL(KID_COMPRESS, k_compress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
  else
    KL(KID_COMPRESS, k_compress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
  HIPCK(hipGetLastError());
  // stable byte-lexicographic sort on the device; the sorted bytes travel to the host for the one sequential SHA-256
  if (!c->ev_host) HIPCK(hipEventCreateWithFlags(&c->ev_host, hipEventDisableTiming));
  bool full_sort = false;
  if ((rc = run_key_sort_to_host(c, d_bytes, n, width, w, c->ev_host, &full_sort))) return rc;
  // H(msg) does not depend on the keys (verify_secure never prefixes them, reference src/secure_aggregation.rs:236-246): the
  // hash-to-curve of the final core_verify -- one wave, 1 ms for G1 and 4 ms for G2 of pure latency -- runs on the side
  // stream beside everything up to the pairing: the key sort, the host's hash of the key stream and the whole key sum; for
  // Bls12381G1Impl the signature's half of the pairing check goes with it (cut_tail_begin).
  uint8_t* d_hash = nullptr;
  CutTail cut;
  const bool use_cut = cut_tail_applies(sig_group, scheme);
  if (use_cut) {
    if ((rc = cut_tail_begin(c, sig_group, scheme, sig, fmt, msg, msg_len, cut))) return rc;
  } else {
    const void* d_msg0;
    if ((rc = stage_in(c, msg, msg_len, &d_msg0))) return rc;
    uint64_t* d_offs0 = (uint64_t*)arena_take(c, 16);
    d_hash = (uint8_t*)arena_take(c, 288);
    if (!d_offs0 || !d_hash) return fail(BLSGPU_E_HIP, "internal: arena too small");
    const uint64_t offs0[2] = {0, (uint64_t)msg_len};
    if ((rc = h2d_small(c, d_offs0, offs0, 16))) return rc;
    if ((rc = side_fork(c))) return rc;
    if (sig_group == 1)
      launch_hash_g1_small(c->side, 1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, 1, scheme_dst(sig_group, scheme), d_hash, (uint32_t*)nullptr);
    else
      launch_hash_g2_small(c->side, 1, (const uint8_t*)d_msg0, (const uint64_t*)d_offs0, 1, scheme_dst(sig_group, scheme), d_hash);
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(c->ev_join, c->side));
  }
  // ... and so does the scalar-independent part of the key sum (every key to affine, with its endomorphism images)
  const bool msm2 = msm_use_pippenger(n) && !msm_use_v1();
  msm2_ws mw;
  if (msm2) {
    if ((rc = msm2_ws_take(c, n, sig_group == 1 ? 2 : 1, mw, msm_tables))) return rc;
    if (sig_group == 1) rc = run_msm2_prep<2>(c, (const uint8_t*)d_pks, fmt, nullptr, n, mw);
    else rc = run_msm2_prep<1>(c, (const uint8_t*)d_pks, fmt, nullptr, n, mw);
    if (rc) return rc;
  }
  if ((rc = run_key_sort_finish(c, d_bytes, n, width, w, c->ev_host, &full_sort))) return rc;
  const double t1 = now();
  uint8_t H[32];
  keys_digest_host(c->hpin, width * n, H);
  const double t2 = now();
  if ((rc = h2d_small(c, d_H, H, 32))) return rc;
  HIPCK(hipMemsetAsync(d_zero, 0, 4, c->stream));
  // t_i on the device, written to the INPUT slot of the key that sorted to position i: the sum below needs no permutation
  if ((rc = run_coefficients(c, d_H, w.perm_a, n, 0, n, 0, d_scal, d_zero))) return rc;
  HIPCK(hipMemcpyAsync(h_zero, d_zero, 4, hipMemcpyDeviceToHost, c->stream));
  // aggregated_pk = sum t_i * pk_sorted[i]   (reference src/secure_aggregation.rs:201-204)
  if (msm2) {
    if (sig_group == 1) rc = run_msm2_rest<2>(c, d_scal, n, mw, d_part, false);
    else rc = run_msm2_rest<1>(c, d_scal, n, mw, d_part, false);
  } else {
    if (sig_group == 1) rc = run_point_sum<2>(c, (const uint8_t*)d_pks, fmt, d_scal, nullptr, n, d_part, T);
    else rc = run_point_sum<1>(c, (const uint8_t*)d_pks, fmt, d_scal, nullptr, n, d_part, T);
  }
  if (use_cut) {
    if ((rc = cut_tail_finish(c, rc, d_part, cut, status))) return rc;
  } else {
    if (rc) return rc;
    HIPCK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    if ((rc = verify_one_tail(c, sig_group, scheme, 0, d_part, sig, fmt, msg, msg_len, status, d_hash))) return rc;
  }
  if (trace)
    fprintf(stderr, "[blsgpu] verify_secure n=%zu: compress + key sort%s + D2H %.2f ms (message hash on the side stream), key-stream SHA-256 on the host %.2f ms, rest %.2f ms\n",
            n, full_sort ? " (full-width)" : "", t1 - t0, t2 - t1, now() - t2);
  if (*h_zero) {                   // a zero coefficient: BlsError::InvalidCoefficient before any verification (:97-100)
    st = BLSGPU_INVALID_COEFFICIENT;
    if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
    else *status = st;
  }
  return 0;
}
API_CATCH

int blsgpu_secure_coefficients(const uint8_t* key_bytes, size_t n, size_t width, uint32_t* out_perm, uint8_t* out_scalars,
                               int32_t* status) try {
  if (!initialised()) return NOT_INIT();
  if (width != 48 && width != 96) return fail(BLSGPU_E_ARG, "width must be 48 or 96");
  if (!status || (n && (!key_bytes || !out_perm || !out_scalars))) return fail(BLSGPU_E_ARG, "null argument");
  if (n >= 0xffffffffull) return fail(BLSGPU_E_ARG, "more than 2^32 - 2 keys");
  int32_t st = BLSGPU_OK;
  if (n) {
    CTX_ACQUIRE(c);
    int rc = arena_reserve(c, pad256(width * n) + pad256(32 * n) + keysort_ws_bytes(n, width) + 4096);
    if (rc) return rc;
    c->arena_off = 0;
    const void* d_kb;
    if ((rc = stage_in(c, key_bytes, width * n, &d_kb))) return rc;
    uint8_t* d_scal = (uint8_t*)arena_take(c, 32 * n);
    uint8_t* d_H = (uint8_t*)arena_take(c, 64);
    int32_t* d_zero = (int32_t*)arena_take(c, 64);
    int32_t* h_zero = (int32_t*)hsmall_take(c, 64);
    keysort_ws w;
    if (!d_scal || !d_H || !d_zero || !h_zero) return fail(BLSGPU_E_HIP, "internal: arena too small");
    if ((rc = keysort_ws_take(c, n, width, w))) return rc;
    if ((rc = run_key_sort_to_host(c, (const uint8_t*)d_kb, n, width, w, nullptr, nullptr))) return rc;
    uint8_t H[32];
    keys_digest_host(c->hpin, width * n, H);
    if ((rc = h2d_small(c, d_H, H, 32))) return rc;
    HIPCK(hipMemsetAsync(d_zero, 0, 4, c->stream));
    if ((rc = run_coefficients(c, d_H, w.perm_a, n, 0, n, 1, d_scal, d_zero))) return rc;
    HIPCK(hipMemcpyAsync(h_zero, d_zero, 4, hipMemcpyDeviceToHost, c->stream));
    if ((rc = copy_out(c, out_perm, w.perm_a, 4 * n))) return rc;
    if ((rc = copy_out(c, out_scalars, d_scal, 32 * n))) return rc;
    SYNC_FLUSH(c);
    if (*h_zero) st = BLSGPU_INVALID_COEFFICIENT;
  }
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  return 0;
}
API_CATCH

/* The three steps of hash_public_keys_with_sorted as separate calls, for callers that shard the keys over several GPUs
 * (agora-blsful_amd/dist.py): every rank sorts the gathered key bytes (cheap, on its device), ONE rank hashes the sorted
 * stream and broadcasts the 32-byte digest, every rank derives the coefficients of its own keys.
 * blsgpu_sort_keys: out_perm[i] = input index of the i-th key in stable byte-lexicographic order. */
int blsgpu_sort_keys(const uint8_t* key_bytes, size_t n, size_t width, uint32_t* out_perm) try {
  if (!initialised()) return NOT_INIT();
  if (width != 48 && width != 96) return fail(BLSGPU_E_ARG, "width must be 48 or 96");
  if (n == 0) return 0;
  if (!key_bytes || !out_perm) return fail(BLSGPU_E_ARG, "null argument");
  if (n >= 0xffffffffull) return fail(BLSGPU_E_ARG, "more than 2^32 - 2 keys");
  CTX_ACQUIRE(c);
  int rc = arena_reserve(c, pad256(width * n) + keysort_ws_bytes(n, width) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void* d_kb;
  if ((rc = stage_in(c, key_bytes, width * n, &d_kb))) return rc;
  keysort_ws w;
  if ((rc = keysort_ws_take(c, n, width, w))) return rc;
  uint32_t* h_flag = (uint32_t*)hsmall_take(c, 64);
  if (!h_flag) return fail(BLSGPU_E_HIP, "internal: pinned record buffer exhausted");
  for (int attempt = 0; attempt < 2; attempt++) {
    if ((rc = run_key_sort_passes(c, (const uint8_t*)d_kb, n, width, attempt == 0 ? keysort_prefix(n) : (int)width, w))) return rc;
    if (attempt == 1) break;
    HIPCK(hipMemsetAsync(w.flag, 0, 4, c->stream));
    KL(KID_KEY_SORT, k_keys_tie_flag, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_kb, width, (size_t)keysort_prefix(n), (const uint32_t*)w.perm_a, w.flag);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(h_flag, w.flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    if (*h_flag == 0) break;
  }
  if ((rc = copy_out(c, out_perm, w.perm_a, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH

/* out_digest[32] = SHA-256 over the keys concatenated in the order of perm (reference src/secure_aggregation.rs:45-59). */
int blsgpu_sorted_keys_digest(const uint8_t* key_bytes, const uint32_t* perm, size_t n, size_t width, uint8_t* out_digest) try {
  if (!initialised()) return NOT_INIT();
  if (width != 48 && width != 96) return fail(BLSGPU_E_ARG, "width must be 48 or 96");
  if (!out_digest || (n && (!key_bytes || !perm))) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
  int rc = arena_reserve(c, 2 * pad256(width * n) + pad256(4 * n) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  if ((rc = pinned_reserve(c, width * n + 64))) return rc;
  if (n) {
    const void *d_kb, *d_perm;
    if ((rc = stage_in(c, key_bytes, width * n, &d_kb))) return rc;
    if ((rc = stage_in(c, perm, 4 * n, &d_perm))) return rc;
    uint8_t* d_sorted = (uint8_t*)arena_take(c, width * n);
    if (!d_sorted) return fail(BLSGPU_E_HIP, "internal: arena too small");
    KL(KID_KEY_SORT, k_keys_gather, dim3(blocks_for(n * (width / 4))), dim3(BLS_BLOCK), n, (const uint8_t*)d_kb, width, (const uint32_t*)d_perm, d_sorted);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(c->hpin, d_sorted, width * n, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
  }
  uint8_t H[32];
  keys_digest_host(c->hpin, width * n, H);
  HIPCK(hipMemcpy(out_digest, H, 32, is_device_ptr(out_digest) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  return 0;
}
API_CATCH

/* t_i = SHA-256(BE32(i) || digest) mod r for the keys of the shard [base, base + count) of the n sorted keys, written in the
 * shard's INPUT order: out_scalars[g - base] belongs to input key g (32 B little-endian each).  *status: BLSGPU_OK or
 * BLSGPU_INVALID_COEFFICIENT (reference src/secure_aggregation.rs:61-100). */
int blsgpu_coefficients_for_range(const uint8_t* digest, const uint32_t* perm, size_t n, size_t base, size_t count,
                                  uint8_t* out_scalars, int32_t* status) try {
  if (!initialised()) return NOT_INIT();
  if (!digest || !status || (n && !perm) || (count && !out_scalars)) return fail(BLSGPU_E_ARG, "null argument");
  if (base + count > n) return fail(BLSGPU_E_ARG, "shard range exceeds the key count");
  int32_t st = BLSGPU_OK;
  if (n) {
    CTX_ACQUIRE(c);
    int rc = arena_reserve(c, pad256(4 * n) + pad256(32 * count) + 4096);
    if (rc) return rc;
    c->arena_off = 0;
    const void *d_perm, *d_H;
    if ((rc = stage_in(c, perm, 4 * n, &d_perm))) return rc;
    if ((rc = stage_in(c, digest, 32, &d_H))) return rc;
    uint8_t* d_scal = is_device_ptr(out_scalars) ? out_scalars : (uint8_t*)arena_take(c, 32 * count);
    int32_t* d_zero = (int32_t*)arena_take(c, 64);
    int32_t* h_zero = (int32_t*)hsmall_take(c, 64);
    if ((count && !d_scal) || !d_zero || !h_zero) return fail(BLSGPU_E_HIP, "internal: arena too small");
    HIPCK(hipMemsetAsync(d_zero, 0, 4, c->stream));
    if ((rc = run_coefficients(c, (const uint8_t*)d_H, (const uint32_t*)d_perm, n, base, count, 0, d_scal, d_zero))) return rc;
    HIPCK(hipMemcpyAsync(h_zero, d_zero, 4, hipMemcpyDeviceToHost, c->stream));
    if (d_scal != out_scalars && (rc = copy_out(c, out_scalars, d_scal, 32 * count))) return rc;
    SYNC_FLUSH(c);
    if (*h_zero) st = BLSGPU_INVALID_COEFFICIENT;
  }
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  return 0;
}
API_CATCH

/* out_idx[p] = input index of the FIRST key equal to the p-th sorted key (perm from blsgpu_sort_keys: stable, so that is
 * the input index at the start of p's run of equal keys) -- the `position` search of aggregate_secure, reference
 * src/secure_aggregation.rs:150-162. */
int blsgpu_first_occurrence(const uint8_t* key_bytes, const uint32_t* perm, size_t n, size_t width, uint32_t* out_idx) try {
  if (!initialised()) return NOT_INIT();
  if (width != 48 && width != 96) return fail(BLSGPU_E_ARG, "width must be 48 or 96");
  if (n == 0) return 0;
  if (!key_bytes || !perm || !out_idx) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
  int rc = arena_reserve(c, pad256(width * n) + 3 * pad256(4 * n) + pad256(4 * scan_tiles(n)) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void *d_kb, *d_perm;
  if ((rc = stage_in(c, key_bytes, width * n, &d_kb))) return rc;
  if ((rc = stage_in(c, perm, 4 * n, &d_perm))) return rc;
  uint32_t* d_start = (uint32_t*)arena_take(c, 4 * n);
  uint32_t* d_tiles = (uint32_t*)arena_take(c, 4 * scan_tiles(n));
  uint32_t* d_idx = is_device_ptr(out_idx) ? out_idx : (uint32_t*)arena_take(c, 4 * n);
  if (!d_start || !d_tiles || !d_idx) return fail(BLSGPU_E_HIP, "internal: arena too small");
  KL(KID_KEY_SORT, k_keys_run_start, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_kb, width, (const uint32_t*)d_perm, d_start);
  if ((rc = run_scan_max_u32(c, KID_KEY_SORT, n, d_start, d_tiles))) return rc;
  KL(KID_KEY_SORT, k_run_first_index, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint32_t*)d_perm, (const uint32_t*)d_start, d_idx);
  HIPCK(hipGetLastError());
  if (d_idx != out_idx && (rc = copy_out(c, out_idx, d_idx, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH

/* Basic's duplicate-message rule on its own (reference src/traits/sig_basic.rs:46-58), for sharded callers that gathered the
 * messages: out2 = (index of the earlier equal message, first index i whose message was seen before), or (~0, ~0). */
int blsgpu_first_duplicate_message(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, uint64_t* out2) try {
  if (!initialised()) return NOT_INIT();
  if (!out2 || !msg_offsets) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
  uint64_t total = 0;
  if (is_device_ptr(msg_offsets)) HIPCK(hipMemcpy(&total, msg_offsets + n, 8, hipMemcpyDeviceToHost));
  else total = msg_offsets[n];
  int rc = arena_reserve(c, pad256(total) + pad256(8 * (n + 1)) + dup_ws_bytes(n) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void *d_msgs, *d_offs;
  if ((rc = stage_in(c, msgs, total, &d_msgs))) return rc;
  if ((rc = stage_in(c, msg_offsets, 8 * (n + 1), &d_offs))) return rc;
  uint64_t* d_out = (uint64_t*)arena_take(c, 64);
  if (!d_out) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if ((rc = run_first_duplicate(c, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, n, d_out))) return rc;
  if ((rc = copy_out(c, out2, d_out, 16))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH

static int hash_to_group(int group, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len, void* out) {
  if (!initialised()) return NOT_INIT();
  if (n == 0) return 0;
  if (!msg_offsets || !out || !dst) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
  // two lanes per message: always for G2; for G1 up to the cooperative threshold (a full hash-only batch is faster with one lane per message)
  const int two = (group == 2 || n <= coop_max_items()) ? 1 : 0;
  if (group == 1 && n <= wide_max_items())   // a few messages: one wave each in the row-wide field type (0.98 ms against 1.6 ms of latency)
  {
    prof_pre(c, KID_HASH);
    launch_hash_g1_small(c->stream, n, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, 0, d, d_out, (uint32_t*)nullptr);
    prof_post(c);
  }
  else if (group == 1) KL(KID_HASH, k_hash_to_g1, dim3(blocks_for(two ? 2 * n : n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, d, d_out, two);
  else if (n <= wide_max_items() && hash_phase_stop() == 0) {   // a few messages: row-wide maps and the engine (one workgroup per message up to 128)
    prof_pre(c, KID_HASH);
    launch_hash_g2_small(c->stream, n, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, 0, d, d_out, n > 128 ? (uint32_t*)arena_take(c, 768 * n) : nullptr);
    prof_post(c);
  } else if (n <= wide_max_items() && hash_phase_stop() != 8) {   // measurement aids (BLSGPU_HASH_STOP=8: clearing in the lane-pair kernel; 9: none)
    KL(KID_HASH, k_hash_to_g2, dim3(blocks_for(2 * n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, d, d_out, 3);
    if (hash_phase_stop() != 9) KL(KID_HASH, k_g2_clear_wide, dim3((unsigned)n), dim3(WIDE_ENGINE_BLOCK), n, d_out);   // 9: measurement aid, no clearing
  } else
    KL(KID_HASH, k_hash_to_g2, dim3(blocks_for(two ? 2 * n : n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_msgs, (const uint64_t*)d_offs, d, d_out, two);
  HIPCK(hipGetLastError());
  if (d_out != out && (rc = copy_out(c, out, d_out, osz * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
int blsgpu_hash_to_g1(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len, void* out) try {
  return hash_to_group(1, msgs, msg_offsets, n, dst, dst_len, out);
}
API_CATCH
int blsgpu_hash_to_g2(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len, void* out) try {
  return hash_to_group(2, msgs, msg_offsets, n, dst, dst_len, out);
}
API_CATCH

static int point_sum_entry(int group, const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out) {
  if (!initialised()) return NOT_INIT();
  if (fmt != BLSGPU_FMT_RAW_PROJ && fmt != BLSGPU_FMT_RAW_AFFINE) return fail(BLSGPU_E_ARG, "fmt must be RAW_PROJ or RAW_AFFINE");
  if (!out || (n && !pts)) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
int blsgpu_msm_g1(const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out) try {
  if (n && !scalars) return fail(BLSGPU_E_ARG, "null scalars");
  return point_sum_entry(1, pts, scalars, n, fmt, out);
}
API_CATCH
int blsgpu_msm_g2(const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out) try {
  if (n && !scalars) return fail(BLSGPU_E_ARG, "null scalars");
  return point_sum_entry(2, pts, scalars, n, fmt, out);
}
API_CATCH

int blsgpu_pairing_product_is_one(const void* g1s, const void* g2s, size_t n, int fmt, int32_t* is_one) try {
  if (!initialised()) return NOT_INIT();
  if (fmt != BLSGPU_FMT_RAW_PROJ && fmt != BLSGPU_FMT_RAW_AFFINE) return fail(BLSGPU_E_ARG, "fmt must be RAW_PROJ or RAW_AFFINE");
  if (!is_one || (n && (!g1s || !g2s))) return fail(BLSGPU_E_ARG, "null argument");
  int32_t verdict = BLSGPU_OK;
  if (n > 0) {
    CTX_ACQUIRE(c);
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
    size_t outputs = 0;
    if ((rc = run_miller_product(c, n, n, d_pairs, d_skip, d_f, &outputs))) return rc;
    if ((rc = run_f12_product_verdict(c, d_f, outputs, n, d_skip + n))) return rc;
    HIPCK(hipMemcpyAsync(&verdict, d_skip + n, 4, hipMemcpyDeviceToHost, c->stream));
    SYNC_FLUSH(c);
  }
  int32_t one = verdict == BLSGPU_OK ? 1 : 0;
  if (is_device_ptr(is_one)) HIPCK(hipMemcpy(is_one, &one, 4, hipMemcpyHostToDevice));
  else *is_one = one;
  return 0;
}
API_CATCH

int blsgpu_serialize(int group, const void* pts, size_t n, int fmt_in, int fmt_out, void* out, int32_t* status) try {
  if (!initialised()) return NOT_INIT();
  if (group != 1 && group != 2) return fail(BLSGPU_E_ARG, "group must be 1 or 2");
  if ((fmt_in != BLSGPU_FMT_RAW_PROJ && fmt_in != BLSGPU_FMT_RAW_AFFINE) || (fmt_out != BLSGPU_FMT_COMPRESSED && fmt_out != BLSGPU_FMT_LEGACY))
    return fail(BLSGPU_E_ARG, "supported conversions: RAW_PROJ/RAW_AFFINE -> COMPRESSED/LEGACY");
  if (n == 0) return 0;
  if (!pts || !out) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
API_CATCH

/* ProofOfPossession::verify for n (pk, proof) pairs: pop_verify(pk, sig) = core_verify(pk, sig, pk.to_bytes(), POP_DST)
 * (reference src/proof_of_possession.rs:79-81, src/traits/sig_pop.rs:67-70; POP_DST src/impls/g1.rs:119, g2.rs:117). */
int blsgpu_pop_verify_batch(int sig_group, const void* pks, const void* proofs, size_t n, int fmt, int32_t* status) try {
  int rc = check_common(sig_group, 0, fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!pks || !proofs || !status) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
  if ((rc = status_out_and_sync(c, status, d_status, n))) return rc;
  return 0;
}
API_CATCH

/* Sign-side secure aggregation: aggregate_secure[_with_mode] / AggregateSignature::from_signatures_secure
 * (reference src/secure_aggregation.rs:110-169,338-352, src/aggregate_signature.rs:191-227): sig_agg = sum t_i * sig[idx_i]
 * over the keys in sorted order, where idx_i is the FIRST input position whose serialised key equals the i-th sorted key
 * (the reference's `position` search, so duplicate keys pick the first matching signature).  n == 0: the identity. */
int blsgpu_aggregate_secure(int sig_group, const void* pks, const void* sigs, size_t n, int ser_format, int fmt, void* out_sig,
                            int32_t* status) try {
  int rc = check_common(sig_group, 0, fmt);
  if (rc) return rc;
  if (!out_sig || !status || (n && (!pks || !sigs))) return fail(BLSGPU_E_ARG, "null argument");
  if (ser_format != 0 && ser_format != 1) return fail(BLSGPU_E_ARG, "ser_format must be 0 (Modern) or 1 (Legacy)");
  if (ser_format == 1 && sig_group != 2) return fail(BLSGPU_E_ARG, "Legacy serialization exists only for Bls12381G2Impl");
  if (n >= 0xffffffffull) return fail(BLSGPU_E_ARG, "more than 2^32 - 2 keys");
  CTX_ACQUIRE(c);
  const size_t psz = pk_size(sig_group, fmt), ssz = sig_size(sig_group, fmt), width = sig_group == 1 ? 96 : 48, T = accumulate_lanes(n);
  const size_t osz = sig_group == 1 ? 144 : 288;
  size_t need = pad256(psz * n) + pad256(ssz * n) + pad256(width * n) + 2 * pad256(4 * n) + pad256(32 * n) + pad256(288 * T) + 8192 + msm_ws_bytes(n) +
                keysort_ws_bytes(n, width) + pad256(4 * scan_tiles(n));
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  int32_t st = BLSGPU_OK;
  uint8_t* d_part = (uint8_t*)arena_take(c, 288 * T);
  int32_t* h_zero = (int32_t*)hsmall_take(c, 64);
  if (!d_part || !h_zero) return fail(BLSGPU_E_HIP, "internal: arena too small");
  *h_zero = 0;
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
    uint32_t* d_start = (uint32_t*)arena_take(c, 4 * n);
    uint32_t* d_tiles = (uint32_t*)arena_take(c, 4 * scan_tiles(n));
    uint8_t* d_scal = (uint8_t*)arena_take(c, 32 * n);
    uint8_t* d_H = (uint8_t*)arena_take(c, 64);
    int32_t* d_zero = (int32_t*)arena_take(c, 64);
    keysort_ws w;
    if (!d_bytes || !d_idx || !d_start || !d_tiles || !d_scal || !d_H || !d_zero) return fail(BLSGPU_E_HIP, "internal: arena too small");
    if ((rc = keysort_ws_take(c, n, width, w))) return rc;
    if (sig_group == 1) KL(KID_COMPRESS, k_compress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
    else KL(KID_COMPRESS, k_compress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, fmt, ser_format, d_bytes);
    HIPCK(hipGetLastError());
    if ((rc = run_key_sort_to_host(c, d_bytes, n, width, w, nullptr, nullptr))) return rc;
    uint8_t H[32];
    keys_digest_host(c->hpin, width * n, H);
    if ((rc = h2d_small(c, d_H, H, 32))) return rc;
    HIPCK(hipMemsetAsync(d_zero, 0, 4, c->stream));
    if ((rc = run_coefficients(c, d_H, w.perm_a, n, 0, n, 1, d_scal, d_zero))) return rc;
    HIPCK(hipMemcpyAsync(h_zero, d_zero, 4, hipMemcpyDeviceToHost, c->stream));
    // the signature of sorted position i is that of the FIRST input position holding an equal key (the reference's
    // `position` search): the sort is stable, so that is the input index at the start of i's run of equal keys
    KL(KID_KEY_SORT, k_keys_run_start, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_bytes, width, (const uint32_t*)w.perm_a, d_start);
    if ((rc = run_scan_max_u32(c, KID_KEY_SORT, n, d_start, d_tiles))) return rc;
    KL(KID_KEY_SORT, k_run_first_index, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint32_t*)w.perm_a, (const uint32_t*)d_start, d_idx);
    HIPCK(hipGetLastError());
    if (sig_group == 1) rc = run_point_sum<1>(c, (const uint8_t*)d_sigs, fmt, d_scal, d_idx, n, d_part, T);
    else rc = run_point_sum<2>(c, (const uint8_t*)d_sigs, fmt, d_scal, d_idx, n, d_part, T);
    if (rc) return rc;
  }
  SYNC_FLUSH(c);
  if (*h_zero) st = BLSGPU_INVALID_COEFFICIENT;
  if (st == BLSGPU_OK) {
    if ((rc = copy_out(c, out_sig, d_part, osz))) return rc;
    SYNC_FLUSH(c);
  }
  if (is_device_ptr(status)) HIPCK(hipMemcpy(status, &st, 4, hipMemcpyHostToDevice));
  else *status = st;
  return 0;
}
API_CATCH

int blsgpu_deserialize(int group, const uint8_t* bytes, size_t n, int fmt_in, void* out, int32_t* status) try {
  if (!initialised()) return NOT_INIT();
  if (group != 1 && group != 2) return fail(BLSGPU_E_ARG, "group must be 1 or 2");
  if (fmt_in != BLSGPU_FMT_COMPRESSED && fmt_in != BLSGPU_FMT_LEGACY) return fail(BLSGPU_E_ARG, "fmt_in must be COMPRESSED or LEGACY");
  if (n == 0) return 0;
  if (!bytes || !out || !status) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
API_CATCH

/* Signature::<C>::try_from(&[u8]) / Vec<u8>::from(&Signature<C>) for n signatures (reference src/signature.rs:112-126): the
 * serde_bare form is the scheme as one byte (enum variant index 0 / 1 / 2) followed by the compressed point, 49 bytes
 * (Bls12381G1Impl) or 97 bytes (Bls12381G2Impl) per record -- the lengths asserted at src/signature.rs:285-286.
 * from_tagged: out_schemes[i] = tag, out = RAW_PROJ points (checked decompression on the device), status[i] = OK or
 * BAD_ENCODING (unknown tag or invalid point; the reference maps every serde error to InvalidInputs(..)). */
int blsgpu_signatures_from_tagged(int sig_group, const uint8_t* bytes, size_t n, uint8_t* out_schemes, void* out, int32_t* status) try {
  int rc = check_common(sig_group, 0, BLSGPU_FMT_RAW_PROJ);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!bytes || !out_schemes || !out || !status) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
  const size_t width = sig_group == 1 ? 48 : 96, osz = sig_group == 1 ? 144 : 288;
  if ((rc = arena_reserve(c, pad256((width + 1) * n) + pad256(width * n) + pad256(n) + pad256(osz * n) + pad256(4 * n) + 4096))) return rc;
  c->arena_off = 0;
  const void* d_in;
  if ((rc = stage_in(c, bytes, (width + 1) * n, &d_in))) return rc;
  uint8_t* d_bytes = (uint8_t*)arena_take(c, width * n);
  uint8_t* d_tags = is_device_ptr(out_schemes) ? out_schemes : (uint8_t*)arena_take(c, n);
  uint8_t* d_out = is_device_ptr(out) ? (uint8_t*)out : (uint8_t*)arena_take(c, osz * n);
  int32_t* d_st = is_device_ptr(status) ? status : (int32_t*)arena_take(c, 4 * n);
  if (!d_bytes || !d_tags || !d_out || !d_st) return fail(BLSGPU_E_HIP, "internal: arena too small");
  KL(KID_DECOMPRESS, k_untag, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, width, (const uint8_t*)d_in, d_bytes, d_tags, d_st);
  HIPCK(hipMemsetAsync(d_out, 0, osz * n, c->stream));      // records with a bad tag are not decoded: leave the identity encoding
  if (sig_group == 1) KL(KID_DECOMPRESS, k_decompress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_bytes, 0, d_out, d_st, 1);
  else KL(KID_DECOMPRESS, k_decompress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_bytes, 0, d_out, d_st, 1);
  HIPCK(hipGetLastError());
  if (d_tags != out_schemes && (rc = copy_out(c, out_schemes, d_tags, n))) return rc;
  if (d_out != out && (rc = copy_out(c, out, d_out, osz * n))) return rc;
  if (d_st != status && (rc = copy_out(c, status, d_st, 4 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH
int blsgpu_signatures_to_tagged(int sig_group, const uint8_t* schemes, const void* sigs, size_t n, int fmt, uint8_t* out) try {
  int rc = check_common(sig_group, 0, fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!schemes || !sigs || !out) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
  const size_t width = sig_group == 1 ? 48 : 96, psz = sig_size(sig_group, fmt);
  if ((rc = arena_reserve(c, pad256(psz * n) + pad256(n) + pad256(width * n) + pad256((width + 1) * n) + 4096))) return rc;
  c->arena_off = 0;
  const void *d_pts, *d_tags;
  if ((rc = stage_in(c, sigs, psz * n, &d_pts))) return rc;
  if ((rc = stage_in(c, schemes, n, &d_tags))) return rc;
  uint8_t* d_bytes = (uint8_t*)arena_take(c, width * n);
  uint8_t* d_out = is_device_ptr(out) ? out : (uint8_t*)arena_take(c, (width + 1) * n);
  if (!d_bytes || !d_out) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if (sig_group == 1) KL(KID_COMPRESS, k_compress<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pts, fmt, 0, d_bytes);
  else KL(KID_COMPRESS, k_compress<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pts, fmt, 0, d_bytes);
  KL(KID_COMPRESS, k_tag, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, width, (const uint8_t*)d_tags, (const uint8_t*)d_bytes, d_out);
  HIPCK(hipGetLastError());
  if (d_out != out && (rc = copy_out(c, out, d_out, (width + 1) * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH

/* Self-test / measurement hook of the row-wide Fp multiplier (csrc/wide.cuh): out[i] = a[i] * b[i]^reps in Fp, 48-byte Montgomery
 * words per element (the Fp coordinate format of RAW_PROJ).  reps > 1 chains the multiplications: with n <= 4 (one wave) the
 * call's device time / reps is the latency of one multiplication of a dependent chain. */
int blsgpu_debug_wide_mul(const uint8_t* a, const uint8_t* b, size_t n, int reps, uint8_t* out) try {
  if (!initialised()) return NOT_INIT();
  if (n == 0) return 0;
  if (!a || !b || !out || reps < 1) return fail(BLSGPU_E_ARG, "bad argument");
  CTX_ACQUIRE(c);
  int rc = arena_reserve(c, 3 * pad256(48 * n) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void *d_a, *d_b;
  if ((rc = stage_in(c, a, 48 * n, &d_a))) return rc;
  if ((rc = stage_in(c, b, 48 * n, &d_b))) return rc;
  uint8_t* d_out = is_device_ptr(out) ? out : (uint8_t*)arena_take(c, 48 * n);
  if (!d_out) return fail(BLSGPU_E_HIP, "internal: arena too small");
  // BLSGPU_WIDE_TEST_BLOCK=256: sixteen items per 256-thread workgroup (the engine's shape) instead of four per wave
  const unsigned tb = knobs().wide_test_block == 256 ? 256 : 64, per = tb / 16;
  KL(KID_WIDE, k_wide_mul_test, dim3((unsigned)((n + per - 1) / per)), dim3(tb), n, (const uint8_t*)d_a, (const uint8_t*)d_b, d_out, reps);
  HIPCK(hipGetLastError());
  if (d_out != out && (rc = copy_out(c, out, d_out, 48 * n))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH

/* Self-test / measurement hook of the row-wide engine (csrc/wide_engine.cuh): runs a caller-supplied program (two words per
 * step, the format of csrc/wide_tables.cuh) `reps` times on one workgroup with F = U = W = ACC = the Fp12 at f_in (twelve
 * 48-byte Montgomery elements, coefficient order of the tables) and T = 0, and returns T.  Steps may only name the Fp12
 * arrays F, T, U, W, ACC. */
int blsgpu_debug_wide_program(const uint32_t* prog, size_t len, int reps, const uint8_t* f_in, uint8_t* t_out) try {
  if (!initialised()) return NOT_INIT();
  if (!prog || !f_in || !t_out || reps < 1 || len < 1 || is_device_ptr(prog)) return fail(BLSGPU_E_ARG, "bad argument");
  if (!wide_prog_is_fp12(prog, len)) return fail(BLSGPU_E_ARG, "program: too long, or a step is not an Fp12 operation on F, T, U, W, ACC");
  CTX_ACQUIRE(c);
  int rc = arena_reserve(c, pad256(8 * len) + 2 * pad256(576) + 4096);
  if (rc) return rc;
  c->arena_off = 0;
  const void *d_prog, *d_in;
  if ((rc = stage_in(c, prog, 8 * len, &d_prog))) return rc;
  if ((rc = stage_in(c, f_in, 576, &d_in))) return rc;
  uint8_t* d_out = is_device_ptr(t_out) ? t_out : (uint8_t*)arena_take(c, 576);
  if (!d_out) return fail(BLSGPU_E_HIP, "internal: arena too small");
  KL(KID_WIDE, k_wide_prog_test, dim3(1), dim3(WIDE_ENGINE_BLOCK), (const uint32_t*)d_prog, (int)len, reps, (const uint8_t*)d_in, d_out);
  HIPCK(hipGetLastError());
  if (d_out != t_out && (rc = copy_out(c, t_out, d_out, 576))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH

int blsgpu_sign_batch(int sig_group, int scheme, const uint8_t* sks, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n,
                      void* out_pks, void* out_sigs) try {
  int rc = check_common(sig_group, scheme, BLSGPU_FMT_RAW_PROJ);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!sks || !msg_offsets || !out_pks || !out_sigs) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
API_CATCH

/* ---- sharded aggregate verify: the local part of core_aggregate_verify (reference
 * src/traits/sig_core.rs:149-178) for a contiguous shard of the (pk, msg) pairs.  out_f12 = product of the Miller
 * values of the shard's pairs [times the (sig, -g) pair when sig != NULL] BEFORE the final exponentiation, as a
 * 576-byte record; *first_bad = local index of the first identity public key, n when the signature is the identity,
 * or -1.  Duplicate-message detection is global and stays with the caller. */
int blsgpu_aggregate_partial(int sig_group, int scheme, const void* pks, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n,
                             const void* sig, int fmt, void* out_f12, int64_t* first_bad) try {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (!out_f12 || !first_bad || !msg_offsets || (n && !pks)) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
  // outputs may be device pointers (the sharded caller hands them to RCCL as they are): nothing crosses to the host then
  uint8_t* d_rec = is_device_ptr(out_f12) ? (uint8_t*)out_f12 : (uint8_t*)arena_take(c, 576);
  int64_t* d_first = is_device_ptr(first_bad) ? first_bad : (int64_t*)arena_take(c, 64);
  if (!d_rec || !d_first) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if ((rc = aggregate_enqueue(c, sig_group, scheme, (const uint8_t*)d_pks, (const uint8_t*)d_sig, fmt, (const uint8_t*)d_msgs,
                              (const uint64_t*)d_offs, n, d_first, nullptr, d_rec)))
    return rc;
  if (d_rec != out_f12 && (rc = copy_out(c, out_f12, d_rec, 576))) return rc;
  if (d_first != first_bad && (rc = copy_out(c, first_bad, d_first, 8))) return rc;
  SYNC_FLUSH(c);
  return 0;
}
API_CATCH

/* BlsSignatureCore::core_verify(pk, sig, msg, dst) with an explicit DST (reference src/traits/sig_core.rs:120-146):
 * n items share one DST; no message augmentation.  Used by the sharded verify_secure tail and by PoP checks
 * (pop_verify = core_verify(pk, sig, pk_bytes, POP_DST), src/traits/sig_pop.rs:67-70). */
static int core_verify_entry(int sig_group, const dst_arg& dst, int aug, const void* pks, const void* sigs, const uint8_t* msgs,
                             const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status) {
  int rc = 0;
  if (n == 0) return 0;
  if (!pks || !sigs || !msg_offsets || !status) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
  if ((rc = status_out_and_sync(c, status, d_status, n))) return rc;
  return 0;
}
int blsgpu_core_verify(int sig_group, const uint8_t* dst, size_t dst_len, const void* pks, const void* sigs, const uint8_t* msgs,
                       const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status) try {
  int rc = check_common(sig_group, 0, fmt);
  if (rc) return rc;
  if (!dst && dst_len) return fail(BLSGPU_E_ARG, "null dst with a non-zero length");
  return core_verify_entry(sig_group, make_dst(dst, dst_len), 0, pks, sigs, msgs, msg_offsets, n, fmt, status);
}
API_CATCH

/* core_verify for n items whose message points H(m_i) are already known (RAW_PROJ, from blsgpu_hash_to_g1/g2 under the
 * scheme's DST): the identity checks of reference src/traits/sig_core.rs:126-135 in the reference's order, then
 * e(H(m), pk) * e(sig, -g) == 1 (:137-145).  Lets a caller hash the message while it is still adding up or exchanging the
 * keys (the sharded MultiSignature::verify / verify_secure of agora-blsful_amd/dist.py do). */
int blsgpu_core_verify_hashed(int sig_group, const void* pks, const void* sigs, const void* hashes, size_t n, int32_t* status) try {
  int rc = check_common(sig_group, 0, BLSGPU_FMT_RAW_PROJ);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!pks || !sigs || !hashes || !status) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
  const size_t pkb = pk_size(sig_group, 0) * n, sgb = sig_size(sig_group, 0) * n;
  size_t need = pad256(pkb) + 2 * pad256(sgb) + pad256(4 * n) + 2 * pad256((size_t)WS_PAIRS_WORDS * 4 * n) + 4096;
  if ((rc = arena_reserve(c, need))) return rc;
  c->arena_off = 0;
  const void *d_pks, *d_sigs, *d_h;
  if ((rc = stage_in(c, pks, pkb, &d_pks))) return rc;
  if ((rc = stage_in(c, sigs, sgb, &d_sigs))) return rc;
  if ((rc = stage_in(c, hashes, sgb, &d_h))) return rc;
  int32_t* d_status = (int32_t*)arena_take(c, 4 * n);
  uint32_t* d_pairs = (uint32_t*)arena_take(c, (size_t)WS_PAIRS_WORDS * 4 * n);
  uint32_t* d_f = (uint32_t*)arena_take(c, (size_t)WS_F_WORDS * 4 * n);
  if (!d_status || !d_pairs || !d_f) return fail(BLSGPU_E_HIP, "internal: arena too small");
  if (sig_group == 1) KL(KID_PREPARE, k_prepare_hashed<1>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_sigs, (const uint8_t*)d_h, d_pairs, d_status, 0);
  else KL(KID_PREPARE, k_prepare_hashed<2>, dim3(blocks_for(n)), dim3(BLS_BLOCK), n, (const uint8_t*)d_pks, (const uint8_t*)d_sigs, (const uint8_t*)d_h, d_pairs, d_status, 0);
  if ((rc = run_pairing2(c, n, d_pairs, d_f, d_status, sig_group == 1 ? 1 : 0))) return rc;
  if ((rc = status_out_and_sync(c, status, d_status, n))) return rc;
  return 0;
}
API_CATCH

/* SignCryptCiphertext::is_valid for n ciphertexts (reference src/sign_crypt_ciphertext.rs:86-101 -> BlsSignCrypt::valid,
 * src/traits/sign_crypt.rs:69-77): W' = H(U.to_bytes() || V) under the scheme's DST, then e(W, -g) * e(W', U) == 1 with
 * U and W not the identity.  That is core_verify(pk := U, sig := W, msg := U.to_bytes() || V, dst): the key-prefixed hash is
 * the augmentation path of k_prepare.  status[i] == 0 <=> valid (Choice 1); any other status <=> Choice 0. */
int blsgpu_signcrypt_valid_batch(int sig_group, int scheme, const void* us, const void* ws, const uint8_t* vs,
                                 const uint64_t* v_offsets, size_t n, int fmt, int32_t* status) try {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  return core_verify_entry(sig_group, scheme_dst(sig_group, scheme), 1, us, ws, vs, v_offsets, n, fmt, status);
}
API_CATCH

/* ProofOfKnowledge::verify for n proofs (reference src/proof_of_knowledge.rs:132-164 -> BlsSignatureProof::verify,
 * src/traits/sig_proof.rs:102-142; verify_timestamp_proof :145-175 derives y and calls the same check): status[i] in
 * {OK, COMMITMENT_IDENTITY, PROOF_IDENTITY, PK_IDENTITY, ZERO_CHALLENGE, INVALID_SIGNATURE (= BlsError::InvalidProof)}.
 * The message is hashed as given under the scheme's DST (the reference does not prefix the key here, even for
 * MessageAugmentation). */
int blsgpu_sig_proof_verify_batch(int sig_group, int scheme, const void* commitments, const void* proofs, const void* pks,
                                  const uint8_t* ys, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, int fmt,
                                  int32_t* status) try {
  int rc = check_common(sig_group, scheme, fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!commitments || !proofs || !pks || !ys || !msg_offsets || !status) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
  if ((rc = status_out_and_sync(c, status, d_status, n))) return rc;
  return 0;
}
API_CATCH

/* n independent two-pair checks  e(g1a[i], g2a[i]) * e(g1b[i], g2b[i]) == 1  -- Pairing::pairing(&[(..), (..)]).is_identity()
 * per item (reference src/traits/pairings.rs:50, glue src/helpers.rs:41-63), the shape of BlsSignCrypt::verify_share
 * (src/traits/sign_crypt.rs:192-207) and the ElGamal / time-lock share checks.  is_one[i] = 1 / 0.  Points must be in the
 * prime-order subgroups (every reference type guarantees it): a pair with an identity member contributes 1. */
int blsgpu_pairing2_check_batch(const void* g1a, const void* g2a, const void* g1b, const void* g2b, size_t n, int fmt,
                                int32_t* is_one) try {
  int rc = check_common(1, 0, fmt);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!g1a || !g2a || !g1b || !g2b || !is_one) return fail(BLSGPU_E_ARG, "null argument");
  CTX_ACQUIRE(c);
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
API_CATCH

/* product of k Fp12 records (the partials of blsgpu_aggregate_partial) -> final exponentiation -> *is_one */
int blsgpu_fp12_product_is_one(const void* f12s, size_t k, int32_t* is_one) try {
  if (!initialised()) return NOT_INIT();
  if (!is_one || (k && !f12s)) return fail(BLSGPU_E_ARG, "null argument");
  int32_t verdict = BLSGPU_OK;
  if (k > 0) {
    CTX_ACQUIRE(c);
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
API_CATCH

}  // extern "C"
