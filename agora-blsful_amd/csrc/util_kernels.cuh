// Byte / index kernels around the arithmetic (included by kernels.cuh; definitions live in the BLS_TU_UTIL translation unit):
//   * multi-workgroup exclusive prefix sum (the MSM's counting sort, the radix sort below)
//   * stable LSD radix sort of fixed-width byte strings -> permutation  (hash_public_keys_with_sorted's sort_by,
//     reference src/secure_aggregation.rs:41-44,273-276: byte-lexicographic and stable)
//   * t_i = SHA-256(BE32(i) || H) mod r per lane                          (reference src/secure_aggregation.rs:61-100)
//   * the Basic scheme's duplicate-message rule on device-resident messages (reference src/traits/sig_basic.rs:46-58)
//   * first-identity-key reduction of aggregate verify                    (reference src/traits/sig_core.rs:155-167)
// All of it is HBM-bound integer/byte work: one wave per workgroup, coalesced where the data allows.
#pragma once

#define UK_SCAN_ITEMS 16                       // counters per lane in a scan tile
#define UK_SCAN_TILE (BLS_BLOCK * UK_SCAN_ITEMS)
#define UK_SORT_ITEMS 16                       // keys per lane in a sort tile
#define UK_SORT_TILE (BLS_BLOCK * UK_SORT_ITEMS)

__global__ void k_scan_tiles(size_t m, const uint32_t* in, uint32_t* out, uint32_t* tile_sums);
__global__ void k_scan_top(size_t ntiles, uint32_t* tile_sums);
__global__ void k_scan_apply(size_t m, uint32_t* out, const uint32_t* tile_sums);
__global__ void k_rs_hist(size_t n, const uint8_t* kb, size_t width, int byte_pos, const uint32_t* perm_in, uint32_t* hist, size_t ntiles);
__global__ void k_rs_scatter(size_t n, const uint8_t* kb, size_t width, int byte_pos, const uint32_t* perm_in, const uint32_t* offs,
                             size_t ntiles, uint32_t* perm_out);
__global__ void k_keys_tie_flag(size_t n, const uint8_t* kb, size_t width, size_t cmp_bytes, const uint32_t* perm, uint32_t* flag);
__global__ void k_keys_gather(size_t n, const uint8_t* kb, size_t width, const uint32_t* perm, uint8_t* out);
__global__ void k_keys_run_start(size_t n, const uint8_t* kb, size_t width, const uint32_t* perm, uint32_t* start);
__global__ void k_run_first_index(size_t n, const uint32_t* perm, const uint32_t* start_scan, uint32_t* idx);
__global__ void k_scan_max_tiles(size_t m, uint32_t* v, uint32_t* tile_max);
__global__ void k_scan_max_top(size_t ntiles, uint32_t* tile_max);
__global__ void k_scan_max_apply(size_t m, uint32_t* v, const uint32_t* tile_max);
__global__ void k_sha256_coeff(size_t n, const uint8_t* H, const uint32_t* perm, size_t base, size_t count, int sorted_order,
                               uint8_t* out_scalars, int32_t* zero_flag);
__global__ void k_iota_u32(size_t n, uint32_t* v);
__global__ void k_first_bad(size_t n, const int32_t* bad, unsigned long long* first);
__global__ void k_first_bad_fin(size_t n, const int32_t* bad, int has_sig, const unsigned long long* first, int64_t* out);
__global__ void k_dup_insert(size_t n, const uint8_t* msgs, const uint64_t* offs, uint32_t mask, uint32_t* tab, uint32_t* minidx,
                             uint32_t* slot_of);
__global__ void k_dup_find(size_t n, const uint32_t* slot_of, const uint32_t* minidx, uint32_t* best);
__global__ void k_dup_fin(const uint32_t* best, const uint32_t* slot_of, const uint32_t* minidx, uint64_t* out2);
__global__ void k_untag(size_t n, size_t width, const uint8_t* in, uint8_t* out_bytes, uint8_t* out_tags, int32_t* status);
__global__ void k_tag(size_t n, size_t width, const uint8_t* tags, const uint8_t* in_bytes, uint8_t* out);
// the items idx[0..cnt) of an array of rows (row_words 32-bit words each) / of a ragged byte array, packed; and statuses back
__global__ void k_gather_rows(size_t cnt, const uint32_t* idx, const uint32_t* src, size_t row_words, uint32_t* dst);
__global__ void k_gather_ragged(size_t cnt, const uint32_t* idx, const uint64_t* offs_src, const uint64_t* offs_dst, const uint8_t* src, uint8_t* dst);
__global__ void k_scatter_i32(size_t cnt, const uint32_t* idx, const int32_t* src, int32_t* dst);

#if defined(BLS_TU_UTIL)
// ---------------------------------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes, one wave per workgroup)
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v = o > v ? o : v;
  }
  return v;
}

// exclusive prefix sum over m counters in three launches: tiles of 1,024 (lane t owns 16 consecutive counters, so every
// lane streams 64 contiguous bytes), one wave over the tile sums (looping for more than 64 tiles), add back
__global__ void __launch_bounds__(BLS_BLOCK) k_scan_tiles(size_t m, const uint32_t* in, uint32_t* out, uint32_t* tile_sums) {
  const size_t base = (size_t)blockIdx.x * UK_SCAN_TILE + (size_t)threadIdx.x * UK_SCAN_ITEMS;
  uint32_t v[UK_SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < UK_SCAN_ITEMS; k++) {
    v[k] = base + k < m ? in[base + k] : 0u;
    s += v[k];
  }
  const uint32_t incl = wave_incl_scan_add(s);
  uint32_t run = incl - s;
#pragma unroll
  for (int k = 0; k < UK_SCAN_ITEMS; k++) {
    if (base + k < m) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 63) tile_sums[blockIdx.x] = incl;
}
__global__ void __launch_bounds__(BLS_BLOCK) k_scan_top(size_t ntiles, uint32_t* tile_sums) {
  uint32_t carry = 0;
  for (size_t b = 0; b < ntiles; b += 64) {
    const size_t j = b + threadIdx.x;
    const uint32_t v = j < ntiles ? tile_sums[j] : 0u;
    const uint32_t incl = wave_incl_scan_add(v);
    if (j < ntiles) tile_sums[j] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
}
__global__ void __launch_bounds__(BLS_BLOCK) k_scan_apply(size_t m, uint32_t* out, const uint32_t* tile_sums) {
  const size_t base = (size_t)blockIdx.x * UK_SCAN_TILE + (size_t)threadIdx.x * UK_SCAN_ITEMS;
  const uint32_t add = tile_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < UK_SCAN_ITEMS; k++)
    if (base + k < m) out[base + k] += add;
}
// inclusive running maximum, in place (same three-launch shape)
__global__ void __launch_bounds__(BLS_BLOCK) k_scan_max_tiles(size_t m, uint32_t* v, uint32_t* tile_max) {
  const size_t base = (size_t)blockIdx.x * UK_SCAN_TILE + (size_t)threadIdx.x * UK_SCAN_ITEMS;
  uint32_t x[UK_SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < UK_SCAN_ITEMS; k++) {
    x[k] = base + k < m ? v[base + k] : 0u;
    s = x[k] > s ? x[k] : s;
    x[k] = s;
  }
  const uint32_t incl = wave_incl_scan_max(s);
  uint32_t prev = __shfl_up(incl, 1, 64);
  if (threadIdx.x == 0) prev = 0;
#pragma unroll
  for (int k = 0; k < UK_SCAN_ITEMS; k++)
    if (base + k < m) v[base + k] = x[k] > prev ? x[k] : prev;
  if (threadIdx.x == 63) tile_max[blockIdx.x] = incl;
}
__global__ void __launch_bounds__(BLS_BLOCK) k_scan_max_top(size_t ntiles, uint32_t* tile_max) {
  uint32_t carry = 0;   // exclusive running maximum of the tile maxima
  for (size_t b = 0; b < ntiles; b += 64) {
    const size_t j = b + threadIdx.x;
    const uint32_t v = j < ntiles ? tile_max[j] : 0u;
    const uint32_t incl = wave_incl_scan_max(v);
    uint32_t prev = __shfl_up(incl, 1, 64);
    if (threadIdx.x == 0) prev = 0;
    if (j < ntiles) tile_max[j] = prev > carry ? prev : carry;
    const uint32_t last = __shfl(incl, 63, 64);
    carry = last > carry ? last : carry;
  }
}
__global__ void __launch_bounds__(BLS_BLOCK) k_scan_max_apply(size_t m, uint32_t* v, const uint32_t* tile_max) {
  const size_t base = (size_t)blockIdx.x * UK_SCAN_TILE + (size_t)threadIdx.x * UK_SCAN_ITEMS;
  const uint32_t lo = tile_max[blockIdx.x];
#pragma unroll
  for (int k = 0; k < UK_SCAN_ITEMS; k++)
    if (base + k < m && v[base + k] < lo) v[base + k] = lo;
}

__global__ void __launch_bounds__(BLS_BLOCK) k_iota_u32(size_t n, uint32_t* v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stable LSD radix sort, one byte per pass.  Position i of the current order holds key perm_in[i] (identity when
// perm_in == nullptr); a tile is 1,024 consecutive positions owned by one wave, visited in 16 groups of 64 so that
// (group, lane) order is position order.  hist is digit-major (hist[d * ntiles + tile]): its exclusive prefix sum is the
// first output slot of every (digit, tile), which is all a stable scatter needs.
__device__ __forceinline__ uint32_t rs_digit(const uint8_t* kb, size_t width, int byte_pos, const uint32_t* perm_in, size_t i) {
  const size_t k = perm_in ? perm_in[i] : i;
  return kb[k * width + byte_pos];
}
__global__ void __launch_bounds__(BLS_BLOCK) k_rs_hist(size_t n, const uint8_t* kb, size_t width, int byte_pos, const uint32_t* perm_in,
                                                    uint32_t* hist, size_t ntiles) {
  __shared__ uint32_t cnt[256];
  for (int d = threadIdx.x; d < 256; d += BLS_BLOCK) cnt[d] = 0;
  __syncthreads();
  const size_t t0 = (size_t)blockIdx.x * UK_SORT_TILE;
  for (int g = 0; g < UK_SORT_ITEMS; g++) {
    const size_t i = t0 + (size_t)g * BLS_BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&cnt[rs_digit(kb, width, byte_pos, perm_in, i)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < 256; d += BLS_BLOCK) hist[(size_t)d * ntiles + blockIdx.x] = cnt[d];
}
__global__ void __launch_bounds__(BLS_BLOCK) k_rs_scatter(size_t n, const uint8_t* kb, size_t width, int byte_pos, const uint32_t* perm_in,
                                                       const uint32_t* offs, size_t ntiles, uint32_t* perm_out) {
  __shared__ uint32_t cur[256];
  for (int d = threadIdx.x; d < 256; d += BLS_BLOCK) cur[d] = offs[(size_t)d * ntiles + blockIdx.x];
  __syncthreads();
  const size_t t0 = (size_t)blockIdx.x * UK_SORT_TILE;
  const uint64_t lt = ((uint64_t)1 << threadIdx.x) - 1;
  for (int g = 0; g < UK_SORT_ITEMS; g++) {
    const size_t i = t0 + (size_t)g * BLS_BLOCK + threadIdx.x;
    const bool valid = i < n;
    const uint32_t key = valid ? (perm_in ? perm_in[i] : (uint32_t)i) : 0u;
    const uint32_t d = valid ? kb[(size_t)key * width + byte_pos] : 0u;
    // lanes holding the same digit (match-any by eight ballots)
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & lt), total = (uint32_t)__popcll(peers);
    uint32_t base = 0;
    if (valid) base = cur[d];
    __syncthreads();
    if (valid) {
      perm_out[base + rank] = key;
      if (rank == total - 1) cur[d] = base + total;   // one lane per digit class advances the cursor
    }
    __syncthreads();
  }
}
// flag |= 1 when two neighbours of the sorted order agree in their first cmp_bytes bytes (the prefix sort left a tie)
__global__ void __launch_bounds__(BLS_BLOCK) k_keys_tie_flag(size_t n, const uint8_t* kb, size_t width, size_t cmp_bytes,
                                                          const uint32_t* perm, uint32_t* flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 || i >= n) return;
  const uint8_t *a = kb + (size_t)perm[i - 1] * width, *b = kb + (size_t)perm[i] * width;
  bool eq = true;
  for (size_t k = 0; k < cmp_bytes; k++) eq = eq && a[k] == b[k];
  if (eq) atomicOr(flag, 1u);
}
// the concatenation of the keys in sorted order (what the reference feeds to SHA-256), 4 bytes per lane access
__global__ void __launch_bounds__(BLS_BLOCK) k_keys_gather(size_t n, const uint8_t* kb, size_t width, const uint32_t* perm, uint8_t* out) {
  const size_t wpk = width / 4, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * wpk) return;
  const size_t i = t / wpk, w = t % wpk;
  ((uint32_t*)out)[t] = ((const uint32_t*)(kb + (size_t)perm[i] * width))[w];
}
// start[i] = i when sorted key i differs from sorted key i - 1 (a run of equal keys begins), else 0
__global__ void __launch_bounds__(BLS_BLOCK) k_keys_run_start(size_t n, const uint8_t* kb, size_t width, const uint32_t* perm, uint32_t* start) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = 0;
  if (i > 0) {
    const uint32_t *a = (const uint32_t*)(kb + (size_t)perm[i - 1] * width), *b = (const uint32_t*)(kb + (size_t)perm[i] * width);
    bool eq = true;
    for (size_t k = 0; k < width / 4; k++) eq = eq && a[k] == b[k];
    if (!eq) s = (uint32_t)i;
  }
  start[i] = s;
}
// after the running maximum: idx[i] = input index of the FIRST key of i's run (stable sort: the smallest input index)
__global__ void __launch_bounds__(BLS_BLOCK) k_run_first_index(size_t n, const uint32_t* perm, const uint32_t* start_scan, uint32_t* idx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = perm[start_scan[i]];
}

// ---------------------------------------------------------------------------------------------------------------------
// t_p = int_BE(SHA-256(BE32(p) || H)) mod r for sorted position p (reference src/secure_aggregation.rs:61-100; the
// reduction is SURVEY 8a A9's finding).  sorted_order != 0: out[p] = t_p.  Otherwise the scalar goes to the INPUT slot of
// the key that sorted to p, restricted to the shard [base, base + count): out[perm[p] - base] = t_p.
__device__ __forceinline__ void u256_mod_r_le(uint32_t v[8]) {
  const uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  for (int round = 0; round < 5; round++) {   // 2^256 < 4.5 r
    uint32_t d[8];
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t s = (uint64_t)v[i] - R[i] - bw;
      d[i] = (uint32_t)s;
      bw = (s >> 63) & 1;
    }
    if (!bw) {
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = d[i];
    }
  }
}
__global__ void __launch_bounds__(BLS_BLOCK) k_sha256_coeff(size_t n, const uint8_t* H, const uint32_t* perm, size_t base, size_t count,
                                                         int sorted_order, uint8_t* out_scalars, int32_t* zero_flag) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  size_t slot = p;
  if (!sorted_order) {
    const size_t g = perm[p];
    if (g < base || g >= base + count) return;
    slot = g - base;
  }
  uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  uint32_t w[16];
  w[0] = (uint32_t)p;                               // BE32(i): the word itself in SHA-256's big-endian word order
#pragma unroll
  for (int k = 0; k < 8; k++) w[1 + k] = ((uint32_t)H[4 * k] << 24) | ((uint32_t)H[4 * k + 1] << 16) | ((uint32_t)H[4 * k + 2] << 8) | H[4 * k + 3];
  w[9] = 0x80000000u;
#pragma unroll
  for (int k = 10; k < 15; k++) w[k] = 0;
  w[15] = 36 * 8;
  sha256_compress(h, w);
  uint32_t v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = h[7 - k];      // big-endian digest -> little-endian 32-bit words
  u256_mod_r_le(v);
  uint32_t nz = 0;
  uint32_t* o = (uint32_t*)(out_scalars + 32 * slot);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    nz |= v[k];
    o[k] = v[k];
  }
  if (!nz) atomicOr((int*)zero_flag, 1);           // reference :97-100: a zero coefficient is InvalidCoefficient
}

// ---------------------------------------------------------------------------------------------------------------------
// aggregate verify: first identity public key.  first must be preset to ~0 (= -1).
__global__ void __launch_bounds__(BLS_BLOCK) k_first_bad(size_t n, const int32_t* bad, unsigned long long* first) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && bad[i]) atomicMin(first, (unsigned long long)i);
}
// the signature's identity check comes first in the reference (sig_core.rs:155-159): then the answer is n
__global__ void __launch_bounds__(BLS_BLOCK) k_first_bad_fin(size_t n, const int32_t* bad, int has_sig, const unsigned long long* first, int64_t* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (has_sig && bad[n]) *out = (int64_t)n;
  else *out = (int64_t)*first;
}

// ---------------------------------------------------------------------------------------------------------------------
// Duplicate messages (Basic scheme, reference src/traits/sig_basic.rs:46-58: a HashMap scan in input order that stops at the
// first i whose message was seen before and reports (index of that earlier message, i)).  Device form: every message claims
// a slot of an open-addressing table by compare-and-swap (the slot's first claimant is the class representative; a later
// claimant compares BYTES with it, so the result is exact whatever the 64-bit hash does), every member lowers the slot's
// minimum index; then the smallest i whose class minimum is below i is the reference's i, and that minimum its `old`.
__device__ __forceinline__ uint64_t msg_hash64(const uint8_t* p, size_t len) {
  uint64_t h = 0x9e3779b97f4a7c15ull ^ (len * 0xff51afd7ed558ccdull);
  for (size_t k = 0; k < len; k++) {
    h = (h ^ p[k]) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  h *= 0xc4ceb9fe1a85ec53ull;
  return h ^ (h >> 32);
}
__global__ void __launch_bounds__(BLS_BLOCK) k_dup_insert(size_t n, const uint8_t* msgs, const uint64_t* offs, uint32_t mask, uint32_t* tab,
                                                       uint32_t* minidx, uint32_t* slot_of) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = msgs + offs[i];
  const size_t len = (size_t)(offs[i + 1] - offs[i]);
  uint32_t s = (uint32_t)msg_hash64(p, len) & mask;
  for (;;) {
    uint32_t rep = atomicCAS(&tab[s], 0xffffffffu, (uint32_t)i);
    if (rep == 0xffffffffu) rep = (uint32_t)i;
    bool eq = rep == (uint32_t)i;
    if (!eq && (size_t)(offs[rep + 1] - offs[rep]) == len) {
      const uint8_t* q = msgs + offs[rep];
      eq = true;
      for (size_t k = 0; k < len; k++) eq = eq && p[k] == q[k];
    }
    if (eq) break;
    s = (s + 1) & mask;
  }
  atomicMin(&minidx[s], (uint32_t)i);
  slot_of[i] = s;
}
__global__ void __launch_bounds__(BLS_BLOCK) k_dup_find(size_t n, const uint32_t* slot_of, const uint32_t* minidx, uint32_t* best) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && minidx[slot_of[i]] < (uint32_t)i) atomicMin(best, (uint32_t)i);
}
__global__ void __launch_bounds__(BLS_BLOCK) k_dup_fin(const uint32_t* best, const uint32_t* slot_of, const uint32_t* minidx, uint64_t* out2) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t b = *best;
  if (b == 0xffffffffu) {
    out2[0] = ~0ull;
    out2[1] = ~0ull;
  } else {
    out2[0] = minidx[slot_of[b]];
    out2[1] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// serde_bare form of Signature<C> (reference src/signature.rs:24-44,112-126): the enum's variant index as one byte
// (0 Basic, 1 MessageAugmentation, 2 ProofOfPossession) followed by the compressed point -- 49 / 97 bytes per record
// (the lengths the reference's test asserts, src/signature.rs:285-286).  An unknown tag fails the record.
__global__ void __launch_bounds__(BLS_BLOCK) k_untag(size_t n, size_t width, const uint8_t* in, uint8_t* out_bytes, uint8_t* out_tags, int32_t* status) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = in + i * (width + 1);
  const uint8_t tag = r[0];
  out_tags[i] = tag;
  status[i] = tag <= 2 ? BLS_OK : BLS_ERR_BAD_ENCODING;
  for (size_t k = 0; k < width; k++) out_bytes[i * width + k] = r[1 + k];
}
__global__ void __launch_bounds__(BLS_BLOCK) k_tag(size_t n, size_t width, const uint8_t* tags, const uint8_t* in_bytes, uint8_t* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t* r = out + i * (width + 1);
  r[0] = tags[i];
  for (size_t k = 0; k < width; k++) r[1 + k] = in_bytes[i * width + k];
}
// ---- gather / scatter of item subsets (the per-item fallback of grouped verification)
__global__ void __launch_bounds__(BLS_BLOCK) k_gather_rows(size_t cnt, const uint32_t* idx, const uint32_t* src, size_t row_words, uint32_t* dst) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cnt * row_words) return;
  const size_t j = t / row_words, w = t % row_words;
  dst[t] = src[(size_t)idx[j] * row_words + w];
}
__global__ void __launch_bounds__(BLS_BLOCK) k_gather_ragged(size_t cnt, const uint32_t* idx, const uint64_t* offs_src, const uint64_t* offs_dst, const uint8_t* src,
                                                          uint8_t* dst) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  const uint64_t a = offs_src[idx[j]], len = offs_src[idx[j] + 1] - a, b = offs_dst[j];
  for (uint64_t k = 0; k < len; k++) dst[b + k] = src[a + k];
}
__global__ void __launch_bounds__(BLS_BLOCK) k_scatter_i32(size_t cnt, const uint32_t* idx, const int32_t* src, int32_t* dst) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < cnt) dst[idx[j]] = src[j];
}
#endif  // BLS_TU_UTIL
