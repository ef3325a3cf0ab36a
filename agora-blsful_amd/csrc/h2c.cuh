// RFC 9380 hash_to_curve for the suites BLS12381G1_XMD:SHA-256_SSWU_RO_ and BLS12381G2_XMD:SHA-256_SSWU_RO_:
// expand_message_xmd(SHA-256) -> hash_to_field -> simplified SWU on the isogenous curve -> isogeny -> add ->
// clear cofactor.  Replaces `G1Projective::hash::<ExpandMsgXmd<Sha256>>` / `G2Projective::hash` of the un-vendored
// backend (reference call sites src/impls/g1.rs:18 and src/impls/g2.rs:16; DST passed in by the scheme traits).
#pragma once
#include "curve.cuh"
#include "tower_split.cuh"

// ------------------------------------------------------------------ SHA-256 (streaming, one lane = one hash)
// The block buffer is addressed with a run-time index, so it lives in scratch: bytes are therefore gathered in a register and
// stored a WORD at a time (one store per four bytes, never a read-modify-write), and the compression copies the sixteen words
// into registers once and runs fully unrolled on them.
struct sha256_ctx {
  uint32_t h[8];
  uint32_t w[16];
  uint32_t cur;     // the bytes of the word being filled
  uint32_t fill;    // bytes in the block
  uint32_t total;   // total bytes (messages < 512 MiB)
};

BLS_CONST uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

BLS_FN uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

BLS_FN void sha256_rounds(uint32_t* h, const uint32_t* wsrc) {
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = wsrc[i];
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
      w[i & 15] = wi;
    }
    uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + SHA256_K[i] + wi;
    uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
BLS_NOINLINE void sha256_compress(uint32_t* h, const uint32_t* wsrc) { sha256_rounds(h, wsrc); }

BLS_FN void sha256_init(sha256_ctx& c) {
  c.h[0] = 0x6a09e667; c.h[1] = 0xbb67ae85; c.h[2] = 0x3c6ef372; c.h[3] = 0xa54ff53a;
  c.h[4] = 0x510e527f; c.h[5] = 0x9b05688c; c.h[6] = 0x1f83d9ab; c.h[7] = 0x5be0cd19;
  c.cur = 0;
  c.fill = 0;
  c.total = 0;
}
BLS_FN void sha256_byte(sha256_ctx& c, uint8_t b) {
  c.cur = (c.cur << 8) | b;
  c.fill++;
  c.total++;
  if ((c.fill & 3u) == 0) {
    c.w[(c.fill >> 2) - 1] = c.cur;
    if (c.fill == 64) {
      sha256_compress(c.h, c.w);
      c.fill = 0;
    }
  }
}
BLS_FN void sha256_update(sha256_ctx& c, const uint8_t* p, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) sha256_byte(c, p[i]);
}
BLS_FN void sha256_final(sha256_ctx& c, uint8_t* out) {
  uint32_t bits_lo = c.total << 3, bits_hi = c.total >> 29;
  sha256_byte(c, 0x80);
  while (c.fill != 56) sha256_byte(c, 0);
  c.w[14] = bits_hi;
  c.w[15] = bits_lo;
  sha256_compress(c.h, c.w);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(c.h[i] >> 24);
    out[4 * i + 1] = (uint8_t)(c.h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(c.h[i] >> 8);
    out[4 * i + 3] = (uint8_t)c.h[i];
  }
}

// ------------------------------------------------------------------ expand_message_xmd + hash_to_field
// msg = pre (optional prefix, the Aug scheme's public-key bytes) || m.  NOUT = 128 (G1) or 256 (G2).  dst_len <= 255.
template <int NOUT>
BLS_FN void expand_message_xmd(uint8_t* out, const uint8_t* pre, uint32_t pre_len, const uint8_t* m, uint32_t m_len,
                               const uint8_t* dst, uint32_t dst_len) {
  sha256_ctx c;
  uint8_t b0[32], bi[32];
  sha256_init(c);
  for (int i = 0; i < 64; i++) sha256_byte(c, 0);
  sha256_update(c, pre, pre_len);
  sha256_update(c, m, m_len);
  sha256_byte(c, (uint8_t)(NOUT >> 8));
  sha256_byte(c, (uint8_t)NOUT);
  sha256_byte(c, 0);
  sha256_update(c, dst, dst_len);
  sha256_byte(c, (uint8_t)dst_len);
  sha256_final(c, b0);
  for (int i = 0; i < 32; i++) bi[i] = 0;
  for (int blk = 1; blk <= NOUT / 32; blk++) {
    sha256_init(c);
    for (int i = 0; i < 32; i++) sha256_byte(c, b0[i] ^ bi[i]);
    sha256_byte(c, (uint8_t)blk);
    sha256_update(c, dst, dst_len);
    sha256_byte(c, (uint8_t)dst_len);
    sha256_final(c, bi);
    for (int i = 0; i < 32; i++) out[32 * (blk - 1) + i] = bi[i];
  }
}

// ---- the same in WORDS, for the lane kernels (round 3).  The byte-streaming form above costs a lone wave 0.36 ms per message where its
// instructions are worth 0.05: every byte of the inputs, of b_0 ^ b_(i-1), of the output and of the 64-byte field inputs is a
// memory access of its own (a private byte array or a generic pointer), waited for one at a time.  Here the chaining values are the
// hash state's words, a block is assembled in sixteen registers (a shift register: the block is complete after sixteen words
// whatever it started with), the compression takes and returns its operands BY VALUE (device: vector arguments in registers), and
// the only byte-granular reads left are the message's (when it is not word-aligned) and the tag's.  Requires the prefix and the
// message to be whole words long (the lengths the signature schemes produce: 32-byte messages, 48- / 96-byte keys); anything else
// takes the byte-streaming form.  Output: NOUT / 4 big-endian words.
#if defined(__HIPCC__)
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
__device__ __noinline__ u32x8_t sha256_compress_v(u32x8_t hv, u32x16_t wv) {
  uint32_t h[8], w[16];
#pragma unroll
  for (int i = 0; i < 8; i++) h[i] = hv[i];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = wv[i];
  sha256_rounds(h, w);
  u32x8_t o;
#pragma unroll
  for (int i = 0; i < 8; i++) o[i] = h[i];
  return o;
}
#endif
struct sha_words {
  uint32_t h[8];
  uint32_t w[16];     // the block being filled: a shift register, never indexed with a run-time value
  uint32_t nw;        // words in it
};
BLS_FN void sha_words_block(sha_words& s) {
#if defined(__HIPCC__)
  u32x8_t hv;
  u32x16_t wv;
#pragma unroll
  for (int i = 0; i < 8; i++) hv[i] = s.h[i];
#pragma unroll
  for (int i = 0; i < 16; i++) wv[i] = s.w[i];
  hv = sha256_compress_v(hv, wv);
#pragma unroll
  for (int i = 0; i < 8; i++) s.h[i] = hv[i];
#else
  sha256_rounds(s.h, s.w);
#endif
}
BLS_FN void sha_words_init(sha_words& s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.nw = 0;
}
BLS_FN void sha_words_feed(sha_words& s, uint32_t word) {
#pragma unroll
  for (int i = 0; i < 15; i++) s.w[i] = s.w[i + 1];
  s.w[15] = word;
  if (++s.nw == 16) {
    sha_words_block(s);
    s.nw = 0;
  }
}
// bytes [j, j + 4) of  head[0 .. nhead) || tag || len(tag) || 0x80 || 0 0 0 ...  as one big-endian word
BLS_FN uint32_t xmd_tail_word(uint32_t j, uint32_t nhead, uint32_t head, const uint8_t* dst, uint32_t dst_len) {
  uint32_t word = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t q = j + k;
    uint32_t b;
    if (q < nhead) b = (head >> (8 * (nhead - 1 - q))) & 0xffu;
    else if (q - nhead < dst_len) b = dst[q - nhead];
    else if (q - nhead == dst_len) b = dst_len;
    else if (q - nhead == dst_len + 1) b = 0x80u;
    else b = 0;
    word = (word << 8) | b;
  }
  return word;
}
// ... and the rest of a message of `before` bytes (already fed, a whole number of words): head, tag, its length, the padding, the bit length
BLS_FN void xmd_finish(sha_words& s, uint32_t before, uint32_t nhead, uint32_t head, const uint8_t* dst, uint32_t dst_len) {
  const uint32_t need = nhead + dst_len + 2, total = before + nhead + dst_len + 1;
  for (uint32_t j = 0; j < need || s.nw != 14; j += 4) sha_words_feed(s, xmd_tail_word(j, nhead, head, dst, dst_len));
  sha_words_feed(s, total >> 29);
  sha_words_feed(s, total << 3);
}
BLS_FN uint32_t be32_at(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
template <int NOUT>
BLS_FN void expand_message_xmd_words(uint32_t* out, const uint8_t* pre, uint32_t pre_len, const uint8_t* m, uint32_t m_len,
                                     const uint8_t* dst, uint32_t dst_len) {
  if ((pre_len | m_len) & 3u) {                      // not whole words: the byte-streaming form, its output repacked
    uint8_t ub[NOUT];
    expand_message_xmd<NOUT>(ub, pre, pre_len, m, m_len, dst, dst_len);
    for (int i = 0; i < NOUT / 4; i++) out[i] = be32_at(ub + 4 * i);
    return;
  }
  sha_words s;
  sha_words_init(s);
  for (int i = 0; i < 16; i++) sha_words_feed(s, 0);            // Z_pad: one block of zeros
  for (uint32_t i = 0; i < pre_len; i += 4) sha_words_feed(s, be32_at(pre + i));
  if (((size_t)m & 3u) == 0) {
    for (uint32_t i = 0; i < m_len; i += 4) {
      const uint32_t v = *(const uint32_t*)(m + i);              // little-endian load of big-endian bytes
      sha_words_feed(s, (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24));
    }
  } else {
    for (uint32_t i = 0; i < m_len; i += 4) sha_words_feed(s, be32_at(m + i));
  }
  xmd_finish(s, 64 + pre_len + m_len, 3, (uint32_t)NOUT << 8, dst, dst_len);   // l_i_b_str (two bytes), I2OSP(0, 1)
  uint32_t b0[8], bi[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    b0[i] = s.h[i];
    bi[i] = 0;
  }
  // the tail of every b_i -- I2OSP(i, 1) || tag || its length || the padding -- differs from one i to the next in its first byte
  // only: its words are put together ONCE (tags of up to 93 bytes; all their byte reads issued together), not once per block
  const int TW = 24;
  const uint32_t need1 = dst_len + 3;
  const bool cached = need1 <= 4 * TW;
  uint32_t T[TW];
  if (cached) {
#pragma unroll
    for (int k = 0; k < TW; k++) T[k] = xmd_tail_word(4 * k, 1, 0, dst, dst_len);
  }
  for (int blk = 1; blk <= NOUT / 32; blk++) {
    sha_words_init(s);
#pragma unroll
    for (int i = 0; i < 8; i++) sha_words_feed(s, b0[i] ^ bi[i]);
    if (cached) {
#pragma unroll
      for (int k = 0; k < TW; k++)
        if (4u * k < need1 || s.nw != 14) sha_words_feed(s, k == 0 ? T[0] | ((uint32_t)blk << 24) : T[k]);
      while (s.nw != 14) sha_words_feed(s, 0);
      sha_words_feed(s, 0);                             // 34 + dst_len bytes: the bit length fits the low word
      sha_words_feed(s, (34 + dst_len) << 3);
    } else {
      xmd_finish(s, 32, 1, (uint32_t)blk, dst, dst_len);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      bi[i] = s.h[i];
      out[8 * (blk - 1) + i] = bi[i];
    }
  }
}
// sixteen big-endian words (64 uniform bytes) -> Fp (Montgomery), as fp_from_be64
BLS_FN void fp_from_be64_words(fp& r, const uint32_t* wbe) {
  fp lo, hi, t, k;
  uint32_t w[12];
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = i < 4 ? wbe[3 - i] : 0;
  fp_set_words(hi, w);
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = wbe[15 - i];
  fp_set_words(lo, w);
  fp_load(k, FP_R2);
  fp_mul(t, k, lo);
  fp_load(k, FP_R2_384);
  fp_mul(hi, k, hi);
  fp_add(t, t, hi);
  fp_reduce(r, t);
}

// 64 big-endian bytes -> Fp (Montgomery): (lo48 + hi16 * 2^384) mod p = REDC(lo48 R^2) + REDC(hi16 2^384 R^2)
BLS_FN void fp_from_be64(fp& r, const uint8_t* b) {
  fp lo, hi, t, k;
  uint32_t w[12];
  for (int i = 0; i < 12; i++) w[i] = 0;
  for (int i = 0; i < 4; i++) {
    const uint8_t* q = b + 4 * (3 - i);
    w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  }
  fp_set_words(hi, w);
  for (int i = 0; i < 12; i++) {
    const uint8_t* q = b + 16 + 4 * (11 - i);
    w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  }
  fp_set_words(lo, w);
  fp_load(k, FP_R2);
  fp_mul(t, k, lo);
  fp_load(k, FP_R2_384);
  fp_mul(hi, k, hi);
  fp_add(t, t, hi);
  fp_reduce(r, t);
}

// ------------------------------------------------------------------ G1: simplified SWU (RFC 9380 F.2, straight line)
// returns x = xn/xd and y on E'1.  F = fp (one lane) or wf (one DPP row, csrc/wide.cuh: the single-item latency path)
template <class F>
BLS_NOINLINE void sswu_g1(F& xn, F& xd, F& y, const F& u) {
  F A, B, Z, tv1, tv2, tv3, tv4, tv5, tv6, x, y1, t;
  fp_load(A, SSWU1_A);
  fp_load(B, SSWU1_B);
  fp_load(Z, SSWU1_Z);
  fp_sqr(tv1, u);
  fp_mul(tv1, Z, tv1);
  fp_sqr(tv2, tv1);
  fp_add(tv2, tv2, tv1);
  fp_one(t);
  fp_add(tv3, tv2, t);
  fp_mul(tv3, B, tv3);
  fp_neg(tv4, tv2);
  fp_cmov(tv4, Z, fp_is_zero(tv2));
  fp_mul(tv4, A, tv4);
  fp_sqr(tv2, tv3);
  fp_sqr(tv6, tv4);
  fp_mul(tv5, A, tv6);
  fp_add(tv2, tv2, tv5);
  fp_mul(tv2, tv2, tv3);
  fp_mul(tv6, tv6, tv4);
  fp_mul(tv5, B, tv6);
  fp_add(tv2, tv2, tv5);
  fp_mul(x, tv1, tv3);
  // sqrt_ratio_3mod4(tv2, tv6)
  F s1, s2, s3, y2, c2;
  fp_sqr(s1, tv6);
  fp_mul(s2, tv2, tv6);
  fp_mul(s1, s1, s2);
  fp_pow(y1, s1, EXP_PM3D4, EXP_PM3D4_BITS);
  fp_mul(y1, y1, s2);
  fp_load(c2, SSWU1_C2);
  fp_mul(y2, y1, c2);
  fp_sqr(s3, y1);
  fp_mul(s3, s3, tv6);
  bool is_qr = fp_eq(s3, tv2);
  // y = tv1 * u * y2-branch value when gx1 is not a square
  fp_mul(y, tv1, u);
  fp_mul(y, y, y2);
  fp_cmov(x, tv3, is_qr);
  fp_cmov(y, y1, is_qr);
  F ny;
  fp_neg(ny, y);
  fp_cmov(y, ny, fp_parity(u) != fp_parity(y));
  xn = x;
  xd = tv4;
}

// The four polynomials of the 11-isogeny E'1 -> E1 in homogenised Horner form, sum_i k_i xn^i xd^(D-i), evaluated IN LOCKSTEP over one
// running power xd^j (round 4): acc_q <- acc_q xn + k_q[D_q - j] xd^j for every polynomial of degree D_q >= j.  The same 15 + 2 x 51
// multiplications as with a table of the sixteen powers, but the table was 896 bytes of scratch per lane read through a run-time index (14 loads in
// front of every second multiplication) -- the largest frame of the k_prepare<1> / k_prepare_agg<1> / k_hash_to_g1 family, whose waves wait
// twice as long as the other kernels' (profiles/r04_pmc_default.json).
BLS_NOINLINE void iso_map_g1(g1_jac& r, const fp& xn, const fp& xd, const fp& y) {
  fp XN, XD, YN, YD, pw, c, t;
  fp_load(XN, ISO1_XNUM[11]);
  fp_load(XD, ISO1_XDEN[10]);
  fp_load(YN, ISO1_YNUM[15]);
  fp_load(YD, ISO1_YDEN[15]);
  pw = xd;
#pragma unroll          // static coefficient addresses: their loads are issued ahead of the multiplications instead of one exposed latency per term
  for (int j = 1; j <= 15; j++) {
    if (j > 1) fp_mul(pw, pw, xd);
    if (j <= 11) {
      fp_mul(XN, XN, xn);
      fp_load(c, ISO1_XNUM[11 - j]);
      fp_mul(t, c, pw);
      fp_add(XN, XN, t);
    }
    if (j <= 10) {
      fp_mul(XD, XD, xn);
      fp_load(c, ISO1_XDEN[10 - j]);
      fp_mul(t, c, pw);
      fp_add(XD, XD, t);
    }
    fp_mul(YN, YN, xn);
    fp_load(c, ISO1_YNUM[15 - j]);
    fp_mul(t, c, pw);
    fp_add(YN, YN, t);
    fp_mul(YD, YD, xn);
    fp_load(c, ISO1_YDEN[15 - j]);
    fp_mul(t, c, pw);
    fp_add(YD, YD, t);
  }
  fp zx, yd2;
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

// lane2 >= 0 (device only, latency mode for single verifications): the item occupies two adjacent lanes, lane2 = 0 / 1 maps
// u0 / u1 and the two points are exchanged by DPP, so the two SSWU maps -- half of the hash -- run side by side; both lanes
// then continue with identical operands (q0 + q1 in that order on both) and produce bit-identical results.
#if defined(__HIPCC__)
__device__ __forceinline__ void fp_lane_swap(fp& r, const fp& a) {
#pragma unroll
  for (int i = 0; i < FP_NL; i++) r.l[i] = dpp_swap(a.l[i]);
}
// One G1 doubling by the TWO lanes of an item, which hold the same point: the seven field products of dbl-2009-l
// (jac_dbl_body) run as FOUR steps instead of seven -- lane 0: A = X^2, F = (3A)^2, YZ, E (D - X3); lane 1: B = Y^2,
// C = B^2, (X + B)^2 -- with the same operation order and reductions; lane 0 finishes the formulas (lane 1 executes the same
// instructions on values it discards) and hands the result back.  The 64 doublings of the cofactor clearing are the
// longest serial chain of a single verification's hash.
__device__ __forceinline__ void jac_dbl_pair(g1_jac& p, bool hi) {
  fp s1, s2, s3, in, a, b, o2, o3, D, E, t, x3, y3, z3, c8;
  fp_sel(in, hi, p.y, p.x);
  fp_sqr(s1, in);                 // lane 0: A = X^2          lane 1: B = Y^2
  fp_dbl(E, s1);
  fp_add(E, E, s1);
  fp_reduce(E, E);                // lane 0: E = 3A
  fp_sel(in, hi, s1, E);
  fp_sqr(s2, in);                 // lane 0: F = E^2          lane 1: C = B^2
  fp_add(t, p.x, s1);             //                          lane 1: X + B
  fp_sel(a, hi, t, p.y);
  fp_sel(b, hi, t, p.z);
  fp_mul(s3, a, b);               // lane 0: Y Z              lane 1: (X + B)^2
  fp_lane_swap(o2, s2);           // lane 0: C
  fp_lane_swap(o3, s3);           // lane 0: (X + B)^2
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
  fp_lane_swap(a, x3);
  fp_lane_swap(b, y3);
  fp_lane_swap(t, z3);
  fp_sel(p.x, hi, a, x3);
  fp_sel(p.y, hi, b, y3);
  fp_sel(p.z, hi, t, z3);
}
// [k] P as jac_mul_u64, the doublings shared by the two lanes (both hold the same P and end with the same result)
__device__ __noinline__ void jac_mul_u64_pair(g1_jac& r, const g1_jac& p, uint64_t k, bool hi) {
  g1_jac acc;
  jac_set_inf(acc);
  for (int i = 63; i >= 0; i--) {
    jac_dbl_pair(acc, hi);
    if ((k >> i) & 1) jac_add_body(acc, acc, p);
  }
  r = acc;
}
#endif
// no_clear: stop before the cofactor clearing -- the point of E1(Fp) whose multiple by h_eff is the hash (verification pairs it
// with -[c] g2 instead of -g2 and needs no clearing: csrc/g2neg_lines.cuh)
BLS_NOINLINE void hash_to_g1(g1_jac& r, const uint8_t* pre, uint32_t pre_len, const uint8_t* m, uint32_t m_len,
                       const uint8_t* dst, uint32_t dst_len, int lane2 = -1, bool no_clear = false) {
  uint32_t ub[32];                 // 128 uniform bytes as big-endian words
  expand_message_xmd_words<128>(ub, pre, pre_len, m, m_len, dst, dst_len);
  fp u0, u1, xn, xd, y;
  g1_jac q0, q1;
#if defined(__HIPCC__)
  if (lane2 >= 0) {
    g1_jac mine, other;
    fp_from_be64_words(u0, ub + 16 * lane2);
    sswu_g1(xn, xd, y, u0);
    iso_map_g1(mine, xn, xd, y);
    fp_lane_swap(other.x, mine.x);
    fp_lane_swap(other.y, mine.y);
    fp_lane_swap(other.z, mine.z);
    q0 = lane2 ? other : mine;
    q1 = lane2 ? mine : other;
  } else
#endif
  {
    (void)lane2;
    fp_from_be64_words(u0, ub);
    fp_from_be64_words(u1, ub + 16);
    sswu_g1(xn, xd, y, u0);
    iso_map_g1(q0, xn, xd, y);
    sswu_g1(xn, xd, y, u1);
    iso_map_g1(q1, xn, xd, y);
  }
  jac_add(q0, q0, q1);
  if (no_clear) {
    r = q0;
    return;
  }
  // clear cofactor: h_eff = 1 - x = 1 + |x|
#if defined(__HIPCC__)
  if (lane2 >= 0) jac_mul_u64_pair(q1, q0, BLS_X_ABS, lane2 != 0);
  else
#endif
    jac_mul_u64(q1, q0, BLS_X_ABS);
  jac_add(r, q1, q0);
}

// ------------------------------------------------------------------ G2: simplified SWU (RFC 9380 6.6.2, generic form)
// Projective form: x1 = xn / xd with xn = B (tv2 + 1), xd = -A tv2 (Z A when tv2 = 0), g(x1) = num / xd^3, so
// y1 = sqrt(num xd) / xd^2; the one Fp inversion for 1 / xd rides in the square root's own inversion (fp2_sqrt_inv).
// When g(x1) is not a square, x2 = (Z u^2) x1 and g(x2) = (Z u^2)^3 g(x1).
// F2 = fp2 (one lane) or wf2 (one DPP row, csrc/wide_fp2.cuh: the latency path of Bls12381G2Impl)
template <class F2>
BLS_NOINLINE void sswu_g2(F2& x, F2& y, const F2& u) {
  typedef decltype(F2::c0) F;
  F2 A, B, Z, tv1, tv2, xn, xd, xd2, xd3, num, w, t, Y;
  fp2_load(A, SSWU2_A);
  fp2_load(B, SSWU2_B);
  fp2_load(Z, SSWU2_Z);
  fp2_sqr(tv1, u);
  fp2_mul(tv1, Z, tv1);              // Z u^2
  fp2_sqr(tv2, tv1);
  fp2_add(tv2, tv2, tv1);
  fp2_reduce(tv2, tv2);
  fp2_one(t);
  fp2_add(t, tv2, t);
  fp2_mul(xn, B, t);                 // B (tv2 + 1)
  fp2_neg(t, tv2);
  fp2_cmov(t, Z, fp2_is_zero(tv2));
  fp2_mul(xd, A, t);
  fp2_sqr(xd2, xd);
  fp2_mul(xd3, xd2, xd);
  fp2_sqr(num, xn);
  fp2_mul(t, A, xd2);
  fp2_add(num, num, t);
  fp2_mul(num, num, xn);
  fp2_mul(t, B, xd3);
  fp2_add(num, num, t);              // xn^3 + A xn xd^2 + B xd^3
  fp2_mul(w, num, xd);
  F nxd, nt, inxd;
  fp_sqr(nxd, xd.c0);
  fp_sqr(nt, xd.c1);
  fp_add(nxd, nxd, nt);
  fp_reduce(nxd, nxd);               // norm(xd), non-zero
  // Branch-free choice between x1 and x2 (lanes of a wave would otherwise run the square root twice):
  // sn = norm(w)^((p+1)/4) is a root of norm(w) when g(x1) is a square and satisfies sn^2 = -norm(w) otherwise; in that
  // case norm(w tv1^3) = norm(w) (norm(Z) norm(u)^2)^3 has the root sn * cZ * norm(u)^3 with the constant cZ^2 = -norm(Z)^3.
  F n, sn, c2, nu, nu3, cz, root;
  fp_sqr(n, w.c0);
  fp_sqr(nt, w.c1);
  fp_add(n, n, nt);
  fp_reduce(n, n);
  fp_pow(sn, n, EXP_PM3D4, EXP_PM3D4_BITS);
  fp_mul(sn, sn, n);
  fp_sqr(c2, sn);
  const bool sq1 = fp_eq(c2, n);
  fp_sqr(nu, u.c0);
  fp_sqr(nt, u.c1);
  fp_add(nu, nu, nt);
  fp_reduce(nu, nu);
  fp_sqr(nu3, nu);
  fp_mul(nu3, nu3, nu);
  fp_load(cz, SSWU2_CZ);
  fp_mul(root, sn, cz);
  fp_mul(root, root, nu3);
  fp_cmov(root, sn, sq1);
  {
    F2 t3, w2, xn2;
    fp2_sqr(t3, tv1);
    fp2_mul(t3, t3, tv1);
    fp2_mul(w2, w, t3);
    fp2_mul(xn2, tv1, xn);
    fp2_cmov(w2, w, sq1);
    fp2_cmov(xn2, xn, sq1);
    w = w2;
    xn = xn2;
  }
  if (fp_is_zero(w.c1)) {            // real w (probability 2^-381): the generic routine handles it
    fp2_sqrt_inv(Y, w, &nxd, &inxd);
  } else {                           // complex method with the root of the norm already in hand (tower.cuh fp2_sqrt_inv)
    F h, tt, s, d, prod, pi;
    fp_load(h, FP_HALF);
    fp_add(tt, w.c0, root);
    fp_mul(tt, tt, h);
    fp_pow(s, tt, EXP_PM3D4, EXP_PM3D4_BITS);
    fp_mul(s, s, tt);
    fp_sqr(c2, s);
    const bool direct = fp_eq(c2, tt);
    fp_dbl(d, s);
    fp_reduce(d, d);
    fp_mul(prod, d, nxd);
    fp_inv(pi, prod);
    fp_mul(inxd, pi, d);
    fp_mul(d, pi, nxd);
    fp_mul(d, w.c1, d);
    Y.c0 = s;
    Y.c1 = d;
    F2 alt;
    alt.c0 = d;
    alt.c1 = s;
    fp2_cmov(Y, alt, !direct);
  }
  F2 xdi;
  fp2_conj(xdi, xd);
  fp2_mul_fp(xdi, xdi, inxd);        // 1 / xd
  fp2_mul(x, xn, xdi);
  fp2_sqr(t, xdi);
  fp2_mul(y, Y, t);
  if (fp2_sgn0(u) != fp2_sgn0(y)) fp2_neg(y, y);
}
template <class F2>
BLS_FN void iso2_poly(F2& r, const uint32_t (*k)[2 * FP_NL], int deg, const F2& x) {
  F2 acc, c;
  fp2_load(acc, k[deg]);
  for (int i = deg - 1; i >= 0; i--) {
    fp2_mul(acc, acc, x);
    fp2_load(c, k[i]);
    fp2_add(acc, acc, c);
  }
  r = acc;
}
BLS_NOINLINE void iso_map_g2(g2_jac& r, const fp2& x, const fp2& y) {
  fp2 XN, XD, YN, YD, t, yd2;
  iso2_poly(XN, ISO2_XNUM, 3, x);
  iso2_poly(XD, ISO2_XDEN, 2, x);
  iso2_poly(YN, ISO2_YNUM, 3, x);
  iso2_poly(YD, ISO2_YDEN, 3, x);
  fp2_mul(r.z, XD, YD);
  fp2_sqr(yd2, YD);
  fp2_mul(t, XN, XD);
  fp2_mul(r.x, t, yd2);
  fp2_sqr(t, XD);
  fp2_mul(t, t, XD);
  fp2_mul(t, t, yd2);
  fp2_mul(t, t, YN);
  fp2_mul(r.y, t, y);
}
// the same map to HOMOGENEOUS coordinates (X : Y : Z) = (XN YD : y YN XD : XD YD), three products instead of nine (the complete
// additions of the engine's point programs take them as they are)
template <class F2>
BLS_FN void iso_map_g2_hom(F2& X, F2& Y, F2& Z, const F2& x, const F2& y) {
  F2 XN, XD, YN, YD, t;
  iso2_poly(XN, ISO2_XNUM, 3, x);
  iso2_poly(XD, ISO2_XDEN, 2, x);
  iso2_poly(YN, ISO2_YNUM, 3, x);
  iso2_poly(YD, ISO2_YDEN, 3, x);
  fp2_mul(X, XN, YD);
  fp2_mul(t, YN, XD);
  fp2_mul(Y, t, y);
  fp2_mul(Z, XD, YD);
}
// no_clear: stop before the cofactor clearing (the caller clears it elsewhere: k_g2_clear_wide)
BLS_NOINLINE void hash_to_g2(g2_jac& r, const uint8_t* pre, uint32_t pre_len, const uint8_t* m, uint32_t m_len,
                       const uint8_t* dst, uint32_t dst_len, int lane2 = -1, bool no_clear = false) {
  uint32_t ub[64];                 // 256 uniform bytes as big-endian words
  expand_message_xmd_words<256>(ub, pre, pre_len, m, m_len, dst, dst_len);
  fp2 u0, u1, x, y;
  g2_jac q0, q1;
#if defined(__HIPCC__)
  if (lane2 >= 0) {          // see hash_to_g1; in addition the point addition and the cofactor clearing -- Fp2 arithmetic -- run
    g2_jac mine, other;      // on the lane-split tower (tower_split.cuh): this lane keeps its component of every coordinate
    fp_from_be64_words(u0.c0, ub + 32 * lane2);
    fp_from_be64_words(u0.c1, ub + 32 * lane2 + 16);
    sswu_g2(x, y, u0);
    iso_map_g2(mine, x, y);
    fp_lane_swap(other.x.c0, mine.x.c0);
    fp_lane_swap(other.x.c1, mine.x.c1);
    fp_lane_swap(other.y.c0, mine.y.c0);
    fp_lane_swap(other.y.c1, mine.y.c1);
    fp_lane_swap(other.z.c0, mine.z.c0);
    fp_lane_swap(other.z.c1, mine.z.c1);
    q0 = lane2 ? other : mine;
    q1 = lane2 ? mine : other;
    jac<hfp2> s0, s1, sum, h;
    s0.x.v = lane2 ? q0.x.c1 : q0.x.c0;
    s0.y.v = lane2 ? q0.y.c1 : q0.y.c0;
    s0.z.v = lane2 ? q0.z.c1 : q0.z.c0;
    s1.x.v = lane2 ? q1.x.c1 : q1.x.c0;
    s1.y.v = lane2 ? q1.y.c1 : q1.y.c0;
    s1.z.v = lane2 ? q1.z.c1 : q1.z.c0;
    jac_add(sum, s0, s1);
    if (no_clear) h = sum;
    else g2_clear_cofactor(h, sum);
    fp px, py, pz;
    fp_lane_swap(px, h.x.v);
    fp_lane_swap(py, h.y.v);
    fp_lane_swap(pz, h.z.v);
    r.x.c0 = lane2 ? px : h.x.v;
    r.x.c1 = lane2 ? h.x.v : px;
    r.y.c0 = lane2 ? py : h.y.v;
    r.y.c1 = lane2 ? h.y.v : py;
    r.z.c0 = lane2 ? pz : h.z.v;
    r.z.c1 = lane2 ? h.z.v : pz;
    return;
  }
#endif
  (void)lane2;
  fp_from_be64_words(u0.c0, ub);
  fp_from_be64_words(u0.c1, ub + 16);
  fp_from_be64_words(u1.c0, ub + 32);
  fp_from_be64_words(u1.c1, ub + 48);
  sswu_g2(x, y, u0);
  iso_map_g2(q0, x, y);
  sswu_g2(x, y, u1);
  iso_map_g2(q1, x, y);
  jac_add(q0, q0, q1);
  if (no_clear) r = q0;
  else g2_clear_cofactor(r, q0);
}
