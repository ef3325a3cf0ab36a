// Host-side SHA-256 for the serial part of the secure-aggregation coefficient step
// (reference src/secure_aggregation.rs:45-59 uses the `sha2` crate): one long hash over the sorted key bytes and
// n 36-byte hashes.  Plain C++, no dependencies.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#define HOST_SHA_X86 1
#endif

#if defined(HOST_SHA_X86)
// SHA-NI path (runtime-dispatched): W_g = msg2(msg1(W_{g-4}, W_{g-3}) + alignr(W_{g-1}, W_{g-2}, 4), W_{g-1})
__attribute__((target("sha,sse4.1,ssse3"))) static inline void host_sha256_blocks_shani(uint32_t h[8], const uint8_t* p, size_t nblk) {
  alignas(16) static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
  __m128i tmp = _mm_loadu_si128((const __m128i*)&h[0]);      // DCBA
  __m128i st1 = _mm_loadu_si128((const __m128i*)&h[4]);      // HGFE
  tmp = _mm_shuffle_epi32(tmp, 0xB1);                        // CDAB
  st1 = _mm_shuffle_epi32(st1, 0x1B);                        // EFGH
  __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);                // ABEF
  st1 = _mm_blend_epi16(st1, tmp, 0xF0);                     // CDGH
  for (size_t b = 0; b < nblk; b++, p += 64) {
    const __m128i save0 = st0, save1 = st1;
    __m128i w[4];
    for (int g = 0; g < 16; g++) {
      __m128i m;
      if (g < 4) {
        m = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * g)), bswap);
      } else {
        m = _mm_sha256msg1_epu32(w[g & 3], w[(g + 1) & 3]);
        m = _mm_add_epi32(m, _mm_alignr_epi8(w[(g + 3) & 3], w[(g + 2) & 3], 4));
        m = _mm_sha256msg2_epu32(m, w[(g + 3) & 3]);
      }
      w[g & 3] = m;
      __m128i x = _mm_add_epi32(m, _mm_load_si128((const __m128i*)&K[4 * g]));
      st1 = _mm_sha256rnds2_epu32(st1, st0, x);
      x = _mm_shuffle_epi32(x, 0x0E);
      st0 = _mm_sha256rnds2_epu32(st0, st1, x);
    }
    st0 = _mm_add_epi32(st0, save0);
    st1 = _mm_add_epi32(st1, save1);
  }
  tmp = _mm_shuffle_epi32(st0, 0x1B);                        // FEBA
  st1 = _mm_shuffle_epi32(st1, 0xB1);                        // DCHG
  st0 = _mm_blend_epi16(tmp, st1, 0xF0);                     // DCBA
  st1 = _mm_alignr_epi8(st1, tmp, 8);                        // HGFE
  _mm_storeu_si128((__m128i*)&h[0], st0);
  _mm_storeu_si128((__m128i*)&h[4], st1);
}
static inline bool host_sha256_have_shani() {
  static const int v = __builtin_cpu_supports("sha") ? 1 : 0;
  return v != 0;
}
#endif

struct host_sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  size_t fill = 0;
  uint64_t total = 0;

  host_sha256() {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h, iv, sizeof h);
  }
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void blocks(const uint8_t* p, size_t nblk) {
#if defined(HOST_SHA_X86)
    if (host_sha256_have_shani()) {
      host_sha256_blocks_shani(h, p, nblk);
      return;
    }
#endif
    for (size_t i = 0; i < nblk; i++) block(p + 64 * i);
  }
  void update(const uint8_t* p, size_t n) {
    total += n;
    while (n) {
      if (fill == 0 && n >= 64) {
        size_t nb = n / 64;
        blocks(p, nb);
        p += 64 * nb;
        n -= 64 * nb;
        continue;
      }
      size_t k = 64 - fill < n ? 64 - fill : n;
      memcpy(buf + fill, p, k);
      fill += k;
      p += k;
      n -= k;
      if (fill == 64) {
        blocks(buf, 1);
        fill = 0;
      }
    }
  }
  void final(uint8_t out[32]) {
    uint64_t bits = total * 8;
    uint8_t pad[72] = {0x80};
    size_t padlen = (fill < 56) ? 56 - fill : 120 - fill;
    uint8_t len[8];
    for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(pad, padlen);
    update(len, 8);
    for (int i = 0; i < 8; i++) {
      out[4 * i] = (uint8_t)(h[i] >> 24);
      out[4 * i + 1] = (uint8_t)(h[i] >> 16);
      out[4 * i + 2] = (uint8_t)(h[i] >> 8);
      out[4 * i + 3] = (uint8_t)h[i];
    }
  }
};
