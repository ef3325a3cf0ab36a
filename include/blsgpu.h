/* blsgpu.h -- C ABI of libblsgpu.so: MI355X-native batch BLS12-381 signature verification.
 *
 * This is the drop-in boundary for the verify path of dashpay/agora-blsful (`blsful` 3.0.0-pre8).  Each entry
 * point names the reference interface it replaces (file:line relative to the reference crate); INTEGRATION.md
 * shows the Rust binding a maintainer adds behind a `hip` cargo feature.
 *
 * Conventions
 *   - Every buffer is caller-owned and borrowed for the duration of the call.  Pointers may be host pointers or
 *     HIP device pointers; the library detects which (hipPointerGetAttributes) and stages host buffers itself.
 *   - All calls are blocking, thread-safe and deterministic.  Each bound device owns a small pool of contexts (stream +
 *     workspace, BLSGPU_CONTEXTS, default 2): concurrent callers lease different contexts and overlap on the device.
 *   - Return value: 0 = the call ran (look at the status outputs), < 0 = runtime failure (HIP error, bad argument);
 *     blsgpu_last_error() then gives a message.  There is NO CPU fallback: without a usable gfx950 device every
 *     compute entry point fails with BLSGPU_E_NO_DEVICE.
 *   - sig_group selects the reference's backend type: 1 = Bls12381G1Impl (signature in G1, public key in G2,
 *     src/impls/g1.rs), 2 = Bls12381G2Impl (signature in G2, public key in G1, src/impls/g2.rs).
 *   - scheme = SignatureSchemes (src/sig_types.rs:6-13): 0 Basic, 1 MessageAugmentation, 2 ProofOfPossession.
 *   - Point formats (fmt):
 *       BLSGPU_FMT_RAW_PROJ    Jacobian (X, Y, Z), Montgomery form, little-endian limbs: the in-memory layout of
 *                              blst_p1 (144 B) / blst_p2 (288 B) that G1Projective / G2Projective wrap; Z = 0 is
 *                              the identity.  A `&[PublicKey<C>]` slice can be passed as it is.  RAW points are what the
 *                              reference's types hold -- members of the prime-order subgroups (from_bytes checks it);
 *                              verification relies on that (its verdict for a point outside its subgroup, which no
 *                              reference value can be, is unspecified).
 *       BLSGPU_FMT_RAW_AFFINE  (x, y) Montgomery, 96 B / 192 B; all-zero = identity.
 *       BLSGPU_FMT_COMPRESSED  ZCash compressed encoding 48 B / 96 B (modern), checked on decode.
 *       BLSGPU_FMT_LEGACY      Dash legacy header variant of the same (src/impls/legacy.rs:9-67).
 *   - Scalars are 32 bytes little-endian.
 *   - Status codes mirror BlsError (src/error.rs:5-55) on this path; the shim maps them back to the exact
 *     variants and strings.
 */
#ifndef BLSGPU_H
#define BLSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLSGPU_FMT_RAW_PROJ 0
#define BLSGPU_FMT_RAW_AFFINE 1
#define BLSGPU_FMT_COMPRESSED 2
#define BLSGPU_FMT_LEGACY 3

#define BLSGPU_SCHEME_BASIC 0
#define BLSGPU_SCHEME_AUG 1
#define BLSGPU_SCHEME_POP 2

/* per-item / per-call status (>= 0) */
#define BLSGPU_OK 0                  /* Ok(()) */
#define BLSGPU_INVALID_SIGNATURE 1   /* BlsError::InvalidSignature            sig_core.rs:144,176; secure_aggregation.rs:193 */
#define BLSGPU_SIG_IDENTITY 2        /* InvalidInputs("signature is the identity point")        sig_core.rs:127,156 */
#define BLSGPU_PK_IDENTITY 3         /* InvalidInputs("public key is the identity point") :132; aggregate form
                                        "public key at {aux0} is the identity point" (1-based)   sig_core.rs:163-166 */
#define BLSGPU_DUPLICATE_MESSAGE 4   /* InvalidInputs("duplicate messages detected at {aux0} and {aux1}") sig_basic.rs:51-55 */
#define BLSGPU_INVALID_COEFFICIENT 5 /* BlsError::InvalidCoefficient           secure_aggregation.rs:99,326 */
#define BLSGPU_BAD_LENGTH 6          /* BlsError::InvalidLength                public_key.rs:159-164 */
#define BLSGPU_BAD_ENCODING 7        /* BlsError::DeserializationError         legacy.rs:76-78,110,121 */
#define BLSGPU_LEGACY_FORMAT 8       /* BlsError::LegacyFormatError            legacy.rs:54-57 */
/* blsgpu_sig_proof_verify_batch only (src/traits/sig_proof.rs:110-128); there BLSGPU_INVALID_SIGNATURE stands for
 * BlsError::InvalidProof (:140) and BLSGPU_PK_IDENTITY for InvalidInputs("pk is the identity point") (:120-124) */
#define BLSGPU_COMMITMENT_IDENTITY 9 /* InvalidInputs("commitment is the identity point")        sig_proof.rs:110-114 */
#define BLSGPU_PROOF_IDENTITY 10     /* InvalidInputs("proof is the identity point")             sig_proof.rs:115-119 */
#define BLSGPU_ZERO_CHALLENGE 11     /* InvalidInputs("y is the zero")                           sig_proof.rs:125-127 */

/* runtime failures (< 0): return codes.  One of them can also appear IN a status entry: BLSGPU_E_HIP when the device-side work of
 * that item failed (single-verdict checks run on workgroups that wait for each other with a bound; a wait that ran out is not a
 * verdict).  Calls whose statuses pass through host memory report it as their return code instead. */
#define BLSGPU_E_NO_DEVICE (-1)
#define BLSGPU_E_HIP (-2)
#define BLSGPU_E_ARG (-3)
#define BLSGPU_E_NOT_INIT (-4)

/* Library life cycle.  device = HIP ordinal, or -1 for the current device.  Idempotent for the same device; a second call
 * that names a DIFFERENT device fails with BLSGPU_E_ARG (shut down first to rebind). */
int blsgpu_init(int device);
/* One process driving several GPUs: binds the first ndev visible devices (ndev <= 0: all) and returns how many are bound
 * (or < 0).  blsgpu_verify_batch, blsgpu_multi_verify, blsgpu_aggregate_verify and blsgpu_verify_secure then shard
 * their items over the bound devices inside the library (contiguous ranges, one host thread per device) and fold the
 * per-device partial results -- key sums, Fp12 Miller products, MSM partials -- on device 0; results are identical to
 * the single-device ones.  The one-process-per-GPU alternative over RCCL is agora-blsful_amd/dist.py. */
int blsgpu_init_devices(int ndev);
int blsgpu_device_count(void);
void blsgpu_shutdown(void);
/* copies the last error message of the calling thread's most recent failing call; returns its length */
size_t blsgpu_last_error(char* buf, size_t cap);

/* NEW additive API (no batch entry exists in the reference: every reference call verifies one signature).
 * status[i] = verdict of  Signature::<C>::verify(&pk[i], msg[i])              src/signature.rs:130-138
 *   -> BlsSignature{Basic,MessageAugmentation,Pop}::verify                    src/traits/sig_basic.rs:36-38,
 *                                                                             sig_aug.rs:20-24, sig_pop.rs:37-39
 *   -> BlsSignatureCore::core_verify                                          src/traits/sig_core.rs:120-146
 * msgs is the concatenation of all messages, msg_offsets has n + 1 entries. */
int blsgpu_verify_batch(int sig_group, int scheme, const void* pks, const void* sigs, const uint8_t* msgs,
                        const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status);

/* BlsSignatureCore::core_verify(pk, sig, msg, dst) for n items that share an explicit DST, no message augmentation
 * (reference src/traits/sig_core.rs:120-146).  pop_verify is core_verify(pk, sig, pk_bytes, POP_DST)
 * (src/traits/sig_pop.rs:67-70); the sharded verify_secure tail uses it with the scheme's DST. */
int blsgpu_core_verify(int sig_group, const uint8_t* dst, size_t dst_len, const void* pks, const void* sigs,
                       const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, int fmt, int32_t* status);

/* core_verify for n items whose message points H(m_i) are already known (RAW_PROJ, from blsgpu_hash_to_g1/g2 under the
 * scheme's DST): the identity checks of src/traits/sig_core.rs:126-135 in the reference's order, then the pairing check
 * (:137-145).  Lets a caller hash the message while it still adds up or exchanges keys.  All points RAW_PROJ. */
int blsgpu_core_verify_hashed(int sig_group, const void* pks, const void* sigs, const void* hashes, size_t n, int32_t* status);

/* MultiSignature::<C>::verify(MultiPublicKey::from_public_keys(pks), msg)     src/multi_signature.rs:127-135,
 * src/multi_public_key.rs:79-83 -> BlsMultiKey::from_public_keys (serial sum) src/traits/pk_multi.rs:7-13;
 * fused form BlsSignaturePop::multi_sig_verify                                src/traits/sig_pop.rs:42-49. */
int blsgpu_multi_verify(int sig_group, int scheme, const void* pks, size_t n, const void* sig, const uint8_t* msg,
                        size_t msg_len, int fmt, int32_t* status);

/* AggregateSignature::<C>::verify(&[(pk, msg)])                               src/aggregate_signature.rs:230-239
 *   -> scheme aggregate_verify (Basic: duplicate-message rejection)           src/traits/sig_basic.rs:41-64,
 *                                                                             sig_aug.rs:27-38, sig_pop.rs:52-58
 *   -> BlsSignatureCore::core_aggregate_verify                                src/traits/sig_core.rs:149-178
 * aux[0], aux[1] receive the indices that the reference formats into its error strings (see status codes). */
int blsgpu_aggregate_verify(int sig_group, int scheme, const void* pks, const uint8_t* msgs,
                            const uint64_t* msg_offsets, size_t n, const void* sig, int fmt, int32_t* status,
                            uint64_t* aux);

/* Signature::<C>::verify_secure(&pks, msg)                                    src/signature.rs:177-197
 * and verify_secure_with_mode(&pks, msg, format)                              src/signature.rs:256-276
 *   -> verify_secure_with_dst_internal                                        src/secure_aggregation.rs:173-208
 *   -> hash_public_keys_with_sorted[_mode]                                    src/secure_aggregation.rs:37-106,269-335
 * ser_format: 0 = Modern, 1 = Legacy (only sig_group 2 has 48-byte keys).  n == 0: Ok iff sig is the identity. */
int blsgpu_verify_secure(int sig_group, int scheme, const void* pks, size_t n, const void* sig, const uint8_t* msg,
                         size_t msg_len, int ser_format, int fmt, int32_t* status);

/* The coefficient step of hash_public_keys_with_sorted on already-serialised keys (width = 48 or 96 bytes each):
 * out_perm[i] = input index of the i-th key in sorted order; out_scalars[i] = t_i (32 B LE).
 * src/secure_aggregation.rs:41-103. */
int blsgpu_secure_coefficients(const uint8_t* key_bytes, size_t n, size_t width, uint32_t* out_perm,
                               uint8_t* out_scalars, int32_t* status);

/* The same step split for callers that shard the keys over several GPUs (agora-blsful_amd/dist.py): every rank sorts the
 * gathered key bytes on its device, ONE rank hashes the sorted stream (the only sequential part) and broadcasts the
 * 32-byte digest, every rank derives the coefficients of its own keys.
 *   blsgpu_sort_keys               out_perm[i] = input index of the i-th key in stable byte-lexicographic order  (:41-44)
 *   blsgpu_sorted_keys_digest      out_digest = SHA-256(keys concatenated in the order of perm)                  (:45-59)
 *   blsgpu_coefficients_for_range  t_i = SHA-256(BE32(i) || digest) mod r for the input keys [base, base + count):
 *                                  out_scalars[g - base] (32 B LE) belongs to input key g; *status OK or
 *                                  INVALID_COEFFICIENT                                                            (:61-100)
 *   blsgpu_first_occurrence        out_idx[p] = input index of the FIRST key equal to the p-th sorted key: the `position`
 *                                  search of aggregate_secure (duplicated keys take their first signature)        (:150-162) */
int blsgpu_sort_keys(const uint8_t* key_bytes, size_t n, size_t width, uint32_t* out_perm);
int blsgpu_sorted_keys_digest(const uint8_t* key_bytes, const uint32_t* perm, size_t n, size_t width, uint8_t* out_digest);
int blsgpu_coefficients_for_range(const uint8_t* digest, const uint32_t* perm, size_t n, size_t base, size_t count,
                                  uint8_t* out_scalars, int32_t* status);
int blsgpu_first_occurrence(const uint8_t* key_bytes, const uint32_t* perm, size_t n, size_t width, uint32_t* out_idx);

/* HashToPoint::hash_to_point(msg, dst)                                        src/traits/hash_to_point.rs:11,
 * impls src/impls/g1.rs:17-19 (group 1) and src/impls/g2.rs:15-17 (group 2).  out: RAW_PROJ points. */
int blsgpu_hash_to_g1(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len,
                      void* out);
int blsgpu_hash_to_g2(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, const uint8_t* dst, size_t dst_len,
                      void* out);

/* BlsSignatureCore::aggregate_public_keys / aggregate_signatures              src/traits/sig_core.rs:38-59
 * (= BlsMultiKey::from_public_keys, src/traits/pk_multi.rs:7-13).  out: one RAW_PROJ point. */
int blsgpu_sum_g1(const void* pts, size_t n, int fmt, void* out);
int blsgpu_sum_g2(const void* pts, size_t n, int fmt, void* out);

/* sum_i scalars[i] * pts[i]: the loop `aggregated_pk += pk.0 * *coeff`        src/secure_aggregation.rs:201-204
 * (and the sign-side loop :163-166).  out: one RAW_PROJ point.  Scalars are taken modulo the group order r (a reference
 * Scalar is always < r; any 256-bit value is accepted and reduced) and the points must lie in the prime-order subgroup,
 * which every reference type guarantees. */
int blsgpu_msm_g1(const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out);
int blsgpu_msm_g2(const void* pts, const uint8_t* scalars, size_t n, int fmt, void* out);

/* Pairing::pairing(&[(a_i, b_i)]).is_identity()                               src/traits/pairings.rs:50,
 * glue src/helpers.rs:41-63 (G1 member first).  *is_one = 1 iff the product of pairings is the identity. */
int blsgpu_pairing_product_is_one(const void* g1s, const void* g2s, size_t n, int fmt, int32_t* is_one);

/* point codec: to_bytes / from_bytes [_with_mode]                             src/public_key.rs:58-74,146-171,
 * src/impls/legacy.rs:85-170.  group = 1 (G1, 48 B) or 2 (G2, 96 B).  status[i]: 0 or BAD_ENCODING/LEGACY_FORMAT. */
int blsgpu_serialize(int group, const void* pts, size_t n, int fmt_in, int fmt_out, void* out, int32_t* status);

/* ProofOfPossession::<C>::verify(pk) for n pairs: pop_verify = core_verify(pk, proof, pk.to_bytes(), POP_DST)
 * (src/proof_of_possession.rs:79-81, src/traits/sig_pop.rs:67-70). */
int blsgpu_pop_verify_batch(int sig_group, const void* pks, const void* proofs, size_t n, int fmt, int32_t* status);

/* aggregate_secure / aggregate_secure_with_mode / AggregateSignature::from_signatures_secure: the signature that
 * verify_secure accepts, sum t_i * sig[idx_i] over the sorted keys (src/secure_aggregation.rs:110-169,338-352,
 * src/aggregate_signature.rs:191-227; duplicate keys pick the first matching signature, as the reference's position
 * search does).  out_sig: one RAW_PROJ point; *status: BLSGPU_OK or BLSGPU_INVALID_COEFFICIENT. */
int blsgpu_aggregate_secure(int sig_group, const void* pks, const void* sigs, size_t n, int ser_format, int fmt,
                            void* out_sig, int32_t* status);

/* OPT-IN grouped verification of independent items (sig_group 1 only): same arguments and status vector as
 * blsgpu_verify_batch (raw formats), but groups of eight items share one final exponentiation through a random linear
 * combination with 128-bit scalars; every group whose combined check fails is re-verified item by item, so
 *   - a valid item is never reported invalid, identity errors are reported as in blsgpu_verify_batch,
 *   - an invalid item is reported valid only if its group's combined check passes.
 * The scalars are derived INSIDE the library from the group's own inputs (SHA-256 over the keys, signatures and messages of
 * the group's eight items, their indices, n and `seed`; csrc/kernels.cuh grouped_scalar): whoever chooses the inputs learns
 * the scalars only once all of them are fixed, so forging takes about 2^128 hash evaluations per group (64-bit scalars, as
 * in round 3, could be ground offline in 2^64 by whoever knows `seed`).  `seed` SHOULD be a fresh random value per call: it makes
 * the scalars unpredictable even to someone who knows every other input.  It needs no secrecy afterwards.  Not the reference's semantics to the last bit (the reference has no batched
 * verification); callers opt in.  Pays on all-valid input; failing groups cost their items' ordinary verification on top. */
int blsgpu_verify_batch_grouped(int sig_group, int scheme, const void* pks, const void* sigs, const uint8_t* msgs, const uint64_t* msg_offsets,
                                size_t n, int fmt, uint64_t seed, int32_t* status);

/* Wire ingest: PublicKey::try_from / from_bytes_with_mode and Signature::from_bytes_with_mode -- checked decompression
 * (on curve, subgroup) of 48-byte (group 1) or 96-byte (group 2) encodings, modern or legacy header
 * (src/public_key.rs:58-74,158-171, src/signature.rs:231-253, src/impls/legacy.rs:39-82,100-126,144-170).
 * out: RAW_PROJ; status[i]: 0, BLSGPU_BAD_ENCODING or BLSGPU_LEGACY_FORMAT.  blsgpu_verify_batch also accepts
 * fmt = BLSGPU_FMT_COMPRESSED / BLSGPU_FMT_LEGACY directly (keys and signatures in the same format); a decode failure
 * becomes that item's status. */
int blsgpu_deserialize(int group, const uint8_t* bytes, size_t n, int fmt_in, void* out, int32_t* status);

/* Signature::<C>::try_from(&[u8]) and Vec<u8>::from(&Signature<C>) (src/signature.rs:112-126): the serde_bare form of the
 * Signature enum is its variant index as one byte (0 Basic, 1 MessageAugmentation, 2 ProofOfPossession) followed by the
 * compressed point -- 49 bytes (Bls12381G1Impl) / 97 bytes (Bls12381G2Impl) per record, the lengths the reference asserts at
 * src/signature.rs:285-286.  n records back to back.  from_tagged: out_schemes[i] = the tag, out = RAW_PROJ points (checked
 * decompression incl. the subgroup test), status[i] = OK or BAD_ENCODING (unknown tag, invalid point; the reference maps
 * every serde error to InvalidInputs(..)). */
int blsgpu_signatures_from_tagged(int sig_group, const uint8_t* bytes, size_t n, uint8_t* out_schemes, void* out, int32_t* status);
int blsgpu_signatures_to_tagged(int sig_group, const uint8_t* schemes, const void* sigs, size_t n, int fmt, uint8_t* out);

/* Sharded aggregate verify (one process per GPU): the shard-local part of core_aggregate_verify
 * (src/traits/sig_core.rs:149-178).  out_f12 (576 B) = an Fp12 value whose final exponentiation is the product of the pairings
 * of the shard's (H(m_i), pk_i) pairs [times (sig, -g) when sig != NULL] -- a product of Miller values (for Bls12381G1Impl
 * taken at the message points before their cofactor clearing, the signature's at -[c] g2, c = h_eff^-1 mod r: the same
 * verdict); only products of such records and their final exponentiation are meaningful -- a record is defined up to factors that
 * the final exponentiation removes (the line values are scaled by elements of Fp2, differently from one library build to another), so
 * compare verdicts, never record bytes, and fold only records that ranks of ONE library build produced; *first_bad = local index of the first identity
 * key, n when the signature is the identity, -1 otherwise (identity pairs contribute 1 to the record, so it can always be
 * folded).  out_f12 and first_bad may be device pointers: then nothing crosses to the host and the caller hands them to
 * RCCL as they are.  Ranks exchange the records (all-gather) and finish with blsgpu_fp12_product_is_one.
 * Duplicate-message detection (Basic, src/traits/sig_basic.rs:46-58) is global: blsgpu_first_duplicate_message. */
int blsgpu_aggregate_partial(int sig_group, int scheme, const void* pks, const uint8_t* msgs, const uint64_t* msg_offsets,
                             size_t n, const void* sig, int fmt, void* out_f12, int64_t* first_bad);
int blsgpu_fp12_product_is_one(const void* f12s, size_t k, int32_t* is_one);
/* The Basic scheme's duplicate-message rule on its own (src/traits/sig_basic.rs:46-58), for sharded callers that gathered
 * the messages: out2 = (index of the earlier equal message, the first index whose message was seen before) -- the two
 * numbers of the reference's error string -- or (~0, ~0) when all messages are distinct.  Exact (bytes are compared). */
int blsgpu_first_duplicate_message(const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, uint64_t* out2);

/* ---- other two-pairing checks of the reference that reuse the same pairing stages ----
 *
 * SignCryptCiphertext::is_valid for n ciphertexts (u, v, w)               src/sign_crypt_ciphertext.rs:86-101
 *   -> BlsSignCrypt::valid: W' = H(u.to_bytes() || v), e(w, -g) * e(W', u) == 1, u and w not the identity
 *                                                                             src/traits/sign_crypt.rs:69-77,153-160
 * vs is the concatenation of the v fields, v_offsets has n + 1 entries.  status[i] == BLSGPU_OK <=> Choice(1); any other
 * status (INVALID_SIGNATURE, SIG_IDENTITY for w, PK_IDENTITY for u) <=> Choice(0). */
int blsgpu_signcrypt_valid_batch(int sig_group, int scheme, const void* us, const void* ws, const uint8_t* vs,
                                 const uint64_t* v_offsets, size_t n, int fmt, int32_t* status);

/* ProofOfKnowledge::<C>::verify(pk, msg, y) for n proofs (u = commitment, v = proof)   src/proof_of_knowledge.rs:132-164
 *   -> BlsSignatureProof::verify                                              src/traits/sig_proof.rs:102-142
 * (ProofOfKnowledgeTimestamp::verify derives y from (u, t) and runs the same check, :145-175 / proof_of_knowledge.rs:287-326).
 * ys: the challenges, 32 B little-endian canonical scalars.  status[i]: OK, COMMITMENT_IDENTITY, PROOF_IDENTITY,
 * PK_IDENTITY, ZERO_CHALLENGE (checked in that order) or INVALID_SIGNATURE (= BlsError::InvalidProof). */
int blsgpu_sig_proof_verify_batch(int sig_group, int scheme, const void* commitments, const void* proofs, const void* pks,
                                  const uint8_t* ys, const uint8_t* msgs, const uint64_t* msg_offsets, size_t n, int fmt,
                                  int32_t* status);

/* n independent products of two pairings: is_one[i] = Pairing::pairing(&[(g1a_i, g2a_i), (g1b_i, g2b_i)]).is_identity()
 * (src/traits/pairings.rs:50, src/helpers.rs:41-63) -- the shape of BlsSignCrypt::verify_share
 * (src/traits/sign_crypt.rs:192-207: pairs (-W', share), (w, pk); its identity checks stay with the caller).
 * Points must lie in the prime-order subgroups, which every reference type guarantees. */
int blsgpu_pairing2_check_batch(const void* g1a, const void* g2a, const void* g1b, const void* g2b, size_t n, int fmt,
                                int32_t* is_one);

/* Measurement hooks (not part of the reference interface): when enabled, every kernel launch of the library is
 * bracketed by HIP events on the library's own stream; blsgpu_profile_get returns the accumulated device time and
 * launch count per kernel since the last enable.  bench.py derives the roofline figures from these. */
int blsgpu_profile_enable(int on);
int blsgpu_profile_count(void);
int blsgpu_profile_get(int kernel_id, char* name, size_t name_cap, double* total_ms, uint64_t* launches);

/* Self-test / measurement hook of the row-wide Fp multiplier behind the single-verification latency path (csrc/wide.cuh):
 * out[i] = a[i] * b[i]^reps in Fp, elements as 48-byte Montgomery words (the coordinate format of RAW_PROJ). */
int blsgpu_debug_wide_mul(const uint8_t* a, const uint8_t* b, size_t n, int reps, uint8_t* out);
/* The same for the table-driven engine on top of it (csrc/wide_engine.cuh): runs `prog` (len steps of two words, the
 * format of csrc/wide_tables.cuh, Fp12 operations on the arrays F, T, U, W, ACC only) `reps` times on one workgroup with
 * F = U = W = ACC = the Fp12 at f_in (twelve 48-byte Montgomery elements) and T = 0; t_out <- T.  prog: host memory. */
int blsgpu_debug_wide_program(const uint32_t* prog, size_t len, int reps, const uint8_t* f_in, uint8_t* t_out);

/* Sign side, provided so that benchmarks and tests can build inputs on the device:
 * pk[i] = sk[i] * g (SecretKey::public_key, src/secret_key.rs:342-344) and
 * sig[i] = sk[i] * H(msg[i]) (BlsSignatureCore::core_sign, src/traits/sig_core.rs:108-117; the Aug scheme prefixes
 * the public-key bytes, src/traits/sig_aug.rs:12-17).  sks: 32 B little-endian each; outputs RAW_PROJ. */
int blsgpu_sign_batch(int sig_group, int scheme, const uint8_t* sks, const uint8_t* msgs, const uint64_t* msg_offsets,
                      size_t n, void* out_pks, void* out_sigs);

#ifdef __cplusplus
}
#endif
#endif /* BLSGPU_H */
