/* ORACLE (test infrastructure), sanitizer leg: the threaded CPU legs of oracle/c on inputs shaped like tools/bench_configs.py's
 * (config 4: 16,384 (key, message) pairs on 16 threads and a 1-thread run; config 3: a key sum; config 5: verify_secure on 16
 * threads; a threaded verify batch), built with -fsanitize=address,undefined by tests/test_oracle_c.py.  Written after the
 * round-2 run of tools/bench_configs.py that dumped core in its config-4 section (gpurun_out/r2b/configs.err) and was never
 * explained: the tree that ran was a minute older than the commit that introduced these legs.  Exit code 0 = no finding and the
 * expected status codes. */
#include "bls381_oracle.c"

static int fails = 0;
#define EXPECT(what, got, want) do { long g_ = (long)(got), w_ = (long)(want); if (g_ != w_) { fprintf(stderr, "san_driver: %s = %ld, expected %ld\n", what, g_, w_); fails++; } } while (0)

int main(int argc, char** argv) {
  size_t n = argc > 1 ? (size_t)atol(argv[1]) : 16384;
  int threads = argc > 2 ? atoi(argv[2]) : 16;
  bo_init();
  /* consecutive keys pk_i = (i + 1) g2 (RAW_PROJ is the in-memory g2p), distinct 32-byte messages */
  g2p* pks = (g2p*)xmalloc(sizeof(g2p) * n);
  uint8_t* msgs = (uint8_t*)xmalloc(32 * n);
  uint64_t* offs = (uint64_t*)xmalloc(8 * (n + 1));
  g2p acc; memset(&acc, 0, sizeof acc);
  for (size_t i = 0; i < n; i++) {
    acc = g2p_add(acc, G2_GEN);
    pks[i] = acc;
    for (int k = 0; k < 32; k++) msgs[32 * i + k] = (uint8_t)(i >> (8 * (k & 3))) ^ (uint8_t)(17 * k);
    offs[i] = 32 * i;
  }
  offs[n] = 32 * n;
  g1p sig = G1_GEN;                       /* not the aggregate: every leg below must answer InvalidSignature (1) after doing all its work */
  uint64_t aux[2];
  /* config 4: Basic (the duplicate rule's sort runs), all threads, then one thread on an eighth of the pairs */
  EXPECT("aggregate_verify, threads", bo_aggregate_verify(1, 0, (const uint8_t*)pks, msgs, offs, n, (const uint8_t*)&sig, threads, aux), 1);
  EXPECT("aggregate_verify, 1 thread", bo_aggregate_verify(1, 0, (const uint8_t*)pks, msgs, offs, n / 8 ? n / 8 : 1, (const uint8_t*)&sig, 1, aux), 1);
  if (n >= 8) {                           /* error precedence: duplicate message, identity signature, first identity key (1-based) */
    uint8_t keep[32]; memcpy(keep, msgs + 32 * 5, 32); memcpy(msgs + 32 * 5, msgs + 32 * 2, 32);
    EXPECT("duplicate rule", bo_aggregate_verify(1, 0, (const uint8_t*)pks, msgs, offs, 8, (const uint8_t*)&sig, threads, aux), 4);
    EXPECT("duplicate indices", aux[0] * 100 + aux[1], 2 * 100 + 5);
    memcpy(msgs + 32 * 5, keep, 32);
    g1p inf1; memset(&inf1, 0, sizeof inf1);
    EXPECT("identity signature", bo_aggregate_verify(1, 0, (const uint8_t*)pks, msgs, offs, 8, (const uint8_t*)&inf1, threads, aux), 2);
    g2p keepk = pks[3]; memset(&pks[3], 0, sizeof(g2p));
    EXPECT("identity key", bo_aggregate_verify(1, 0, (const uint8_t*)pks, msgs, offs, 8, (const uint8_t*)&sig, threads, aux), 3);
    EXPECT("identity key index", aux[0], 4);
    pks[3] = keepk;
  }
  /* config 3: the key sum over all keys, threaded and serial */
  EXPECT("multi_verify, threads", bo_multi_verify(1, 2, (const uint8_t*)pks, n, (const uint8_t*)&sig, msgs, 32, threads), 1);
  EXPECT("multi_verify, 1 thread", bo_multi_verify(1, 2, (const uint8_t*)pks, n, (const uint8_t*)&sig, msgs, 32, 1), 1);
  /* config 5: verify_secure, both key groups, legacy transcode, threaded and serial (a 255-bit scalar multiplication per key: few keys) */
  size_t n5 = n < 64 ? n : 64;
  EXPECT("verify_secure_mt G1Impl", bo_verify_secure_mt(1, 0, (const uint8_t*)pks, n5, (const uint8_t*)&sig, msgs, 32, 0, threads), 1);
  EXPECT("verify_secure G1Impl", bo_verify_secure(1, 0, (const uint8_t*)pks, n5 / 4 ? n5 / 4 : 1, (const uint8_t*)&sig, msgs, 32, 0), 1);
  g1p* pk1 = (g1p*)xmalloc(sizeof(g1p) * n5);
  g1p a1; memset(&a1, 0, sizeof a1);
  for (size_t i = 0; i < n5; i++) { a1 = g1p_add(a1, G1_GEN); pk1[i] = a1; }
  g2p sig2 = G2_GEN;
  EXPECT("verify_secure_mt G2Impl legacy", bo_verify_secure_mt(2, 0, (const uint8_t*)pk1, n5, (const uint8_t*)&sig2, msgs, 32, 1, threads), 1);
  /* config 2's CPU leg: a threaded verify batch */
  size_t nb = n < 32 ? n : 32;
  g1p* sigs = (g1p*)xmalloc(sizeof(g1p) * nb);
  for (size_t i = 0; i < nb; i++) sigs[i] = G1_GEN;
  int32_t* st = (int32_t*)xmalloc(4 * nb);
  bo_verify_batch(1, 2, (const uint8_t*)pks, (const uint8_t*)sigs, msgs, offs, nb, st, threads);
  for (size_t i = 0; i < nb; i++) EXPECT("verify_batch status", st[i], 1);
  free(pks); free(msgs); free(offs); free(pk1); free(sigs); free(st);
  if (!fails) printf("san_driver ok: n=%zu threads=%d\n", n, threads);
  return fails ? 1 : 0;
}
