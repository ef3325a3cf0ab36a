// blsful_hip.hpp -- host-side mirror of the reference's verify-path types above the C ABI of libblsgpu (blsgpu.h).
//
// The reference (dashpay/agora-blsful) is Rust and this image has no Rust toolchain, so the host side that a
// `blsful` maintainer would write in src/hip.rs is written here in C++ with the SAME names, argument
// meaning and error behaviour, so that tests/cpp/mirror_test.cpp reads like the reference's own tests:
//
//   SignatureSchemes, SerializationFormat          src/sig_types.rs:6-13, src/serialization.rs:11-17
//   BlsError / BlsResult                           src/error.rs:5-58
//   SecretKey<C> (test inputs only)                src/secret_key.rs:255-265,342-344
//   PublicKey<C>                                   src/public_key.rs:5-11,58-74,146-171
//   Signature<C>                                   src/signature.rs:25-44,112-126,130-138,177-197,231-276
//   MultiPublicKey<C>, MultiSignature<C>           src/multi_public_key.rs:79-83, src/multi_signature.rs:127-135
//   AggregateSignature<C>                          src/aggregate_signature.rs:191-239
//   ProofOfPossession<C>                           src/proof_of_possession.rs:79-81
//   PublicKeyShare<C>, SignatureShare<C>           src/public_key_share.rs:53-72 (verify = the scheme's verify on the share values)
//   verify_batch (additive; no reference entry)    n independent Signature::verify calls in one launch
//
// Header-only; link with -lblsgpu.  Every call goes to the GPU: there is no CPU path behind these types, and a missing
// device surfaces as BlsError{Kind::Runtime} from the first call.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "blsgpu.h"

namespace blsful {

enum class SignatureSchemes { Basic = 0, MessageAugmentation = 1, ProofOfPossession = 2 };
enum class SerializationFormat { Modern = 0, Legacy = 1 };
using Bytes = std::vector<uint8_t>;

struct BlsError {
  enum class Kind { InvalidSignature, InvalidSignatureScheme, InvalidInputs, InvalidCoefficient, InvalidLength, DeserializationError,
                    LegacyFormatError, InvalidProof, Runtime };
  Kind kind;
  std::string message;   // the reference's string for InvalidInputs; the library's message for Runtime
  bool operator==(const BlsError& o) const { return kind == o.kind && message == o.message; }
};

struct Unit {};
template <class T>
class BlsResult {
 public:
  BlsResult(T v) : v_(std::move(v)) {}
  BlsResult(BlsError e) : v_(std::move(e)) {}
  bool is_ok() const { return v_.index() == 0; }
  bool is_err() const { return !is_ok(); }
  const T& unwrap() const { return std::get<0>(v_); }
  const BlsError& unwrap_err() const { return std::get<1>(v_); }

 private:
  std::variant<T, BlsError> v_;
};

namespace detail {
inline BlsError runtime_error(int rc) {
  char buf[512];
  blsgpu_last_error(buf, sizeof buf);
  return BlsError{BlsError::Kind::Runtime, "libblsgpu rc=" + std::to_string(rc) + ": " + buf};
}
// one-time library initialisation on the current device; the error (no gfx950 device, ...) is kept and returned by every call
inline int ensure_init() {
  static int rc = blsgpu_init(-1);
  return rc;
}
// status code -> the exact BlsError value the reference returns (strings: src/traits/sig_core.rs:126-176,
// src/traits/sig_basic.rs:51-55; kinds: src/error.rs:5-55)
inline BlsResult<Unit> from_status(int32_t st, const uint64_t* aux, bool aggregate) {
  using K = BlsError::Kind;
  static const uint64_t none[2] = {0, 0};
  if (!aux) aux = none;
  switch (st) {
    case BLSGPU_OK: return Unit{};
    case BLSGPU_INVALID_SIGNATURE: return BlsError{K::InvalidSignature, ""};
    case BLSGPU_SIG_IDENTITY: return BlsError{K::InvalidInputs, "signature is the identity point"};
    case BLSGPU_PK_IDENTITY:
      if (aggregate) return BlsError{K::InvalidInputs, "public key at " + std::to_string(aux[0]) + " is the identity point"};
      return BlsError{K::InvalidInputs, "public key is the identity point"};
    case BLSGPU_DUPLICATE_MESSAGE:
      return BlsError{K::InvalidInputs, "duplicate messages detected at " + std::to_string(aux[0]) + " and " + std::to_string(aux[1])};
    case BLSGPU_INVALID_COEFFICIENT: return BlsError{K::InvalidCoefficient, ""};
    case BLSGPU_BAD_LENGTH: return BlsError{K::InvalidLength, ""};
    case BLSGPU_BAD_ENCODING: return BlsError{K::DeserializationError, ""};
    case BLSGPU_LEGACY_FORMAT: return BlsError{K::LegacyFormatError, ""};
    default: return BlsError{K::Runtime, "unexpected status " + std::to_string(st)};
  }
}
}  // namespace detail

// ---- the two concrete implementations (reference src/impls/g1.rs, src/impls/g2.rs)
struct Bls12381G1Impl {   // signatures in G1 (48 B), public keys in G2 (96 B)
  static constexpr int SIG_GROUP = 1, PK_GROUP = 2;
  static constexpr size_t PK_RAW = 288, SIG_RAW = 144, PK_BYTES = 96, SIG_BYTES = 48;
};
struct Bls12381G2Impl {   // the Dash orientation: signatures in G2 (96 B), public keys in G1 (48 B)
  static constexpr int SIG_GROUP = 2, PK_GROUP = 1;
  static constexpr size_t PK_RAW = 144, SIG_RAW = 288, PK_BYTES = 48, SIG_BYTES = 96;
};

namespace detail {
// checked decode of one wire encoding into RAW_PROJ (PublicKey::try_from / from_bytes_with_mode, Signature::from_bytes_with_mode)
template <size_t RAW>
BlsResult<std::array<uint8_t, RAW>> decode(int group, const uint8_t* bytes, size_t len, size_t expected, SerializationFormat format) {
  if (int rc = ensure_init()) return runtime_error(rc);
  if (len != expected) return BlsError{BlsError::Kind::InvalidLength, ""};      // src/public_key.rs:159-164, src/signature.rs:236-241
  std::array<uint8_t, RAW> out;
  int32_t st = 0;
  int rc = blsgpu_deserialize(group, bytes, 1, format == SerializationFormat::Legacy ? BLSGPU_FMT_LEGACY : BLSGPU_FMT_COMPRESSED, out.data(), &st);
  if (rc) return runtime_error(rc);
  if (st != BLSGPU_OK) return from_status(st, nullptr, false).unwrap_err();
  return out;
}
template <size_t RAW>
Bytes encode(int group, const std::array<uint8_t, RAW>& raw, size_t width, SerializationFormat format) {
  Bytes out(width);
  int32_t st = 0;
  if (ensure_init() == 0)
    blsgpu_serialize(group, raw.data(), 1, BLSGPU_FMT_RAW_PROJ, format == SerializationFormat::Legacy ? BLSGPU_FMT_LEGACY : BLSGPU_FMT_COMPRESSED,
                     out.data(), &st);
  return out;
}
template <size_t RAW>
BlsResult<std::array<uint8_t, RAW>> sum(int group, const uint8_t* pts, size_t n) {
  if (int rc = ensure_init()) return runtime_error(rc);
  std::array<uint8_t, RAW> out;
  int rc = group == 1 ? blsgpu_sum_g1(pts, n, BLSGPU_FMT_RAW_PROJ, out.data()) : blsgpu_sum_g2(pts, n, BLSGPU_FMT_RAW_PROJ, out.data());
  if (rc) return runtime_error(rc);
  return out;
}
}  // namespace detail

template <class C>
struct PublicKey {
  std::array<uint8_t, C::PK_RAW> raw;   // C::PublicKey as it sits in memory (blst_p1 / blst_p2): what crosses the C ABI

  // PublicKey::try_from(&[u8]), src/public_key.rs:58-74 (Modern encoding, subgroup-checked)
  static BlsResult<PublicKey> try_from(const Bytes& b) { return from_bytes_with_mode(b, SerializationFormat::Modern); }
  // src/public_key.rs:158-171
  static BlsResult<PublicKey> from_bytes_with_mode(const Bytes& b, SerializationFormat format) {
    auto r = detail::decode<C::PK_RAW>(C::PK_GROUP, b.data(), b.size(), C::PK_BYTES, format);
    if (r.is_err()) return r.unwrap_err();
    return PublicKey{r.unwrap()};
  }
  Bytes to_bytes() const { return to_bytes_with_mode(SerializationFormat::Modern); }                      // src/public_key.rs:177-179
  Bytes to_bytes_with_mode(SerializationFormat f) const { return detail::encode(C::PK_GROUP, raw, C::PK_BYTES, f); }   // :146-151
  bool operator==(const PublicKey& o) const { return to_bytes() == o.to_bytes(); }
};

template <class C>
struct Signature;

// test inputs only (the sign side is outside the accelerated path): SecretKey::try_from(32 B big-endian), public_key(), sign()
template <class C>
struct SecretKey {
  std::array<uint8_t, 32> le;   // the scalar, little-endian, as blsgpu_sign_batch takes it
  static BlsResult<SecretKey> try_from(const Bytes& be) {                   // src/secret_key.rs:255-265 (from_be_bytes)
    if (be.size() != 32) return BlsError{BlsError::Kind::InvalidLength, ""};
    SecretKey s;
    for (int i = 0; i < 32; i++) s.le[i] = be[31 - i];
    return s;
  }
  BlsResult<std::pair<PublicKey<C>, Signature<C>>> sign_with_key(SignatureSchemes scheme, const Bytes& msg) const;
  BlsResult<PublicKey<C>> public_key() const {                              // src/secret_key.rs:342-344
    auto r = sign_with_key(SignatureSchemes::Basic, Bytes{});
    if (r.is_err()) return r.unwrap_err();
    return r.unwrap().first;
  }
  BlsResult<Signature<C>> sign(SignatureSchemes scheme, const Bytes& msg) const {   // src/secret_key.rs:347-371
    auto r = sign_with_key(scheme, msg);
    if (r.is_err()) return r.unwrap_err();
    return r.unwrap().second;
  }
};

template <class C>
struct Signature {
  SignatureSchemes scheme;                 // the enum tag of src/signature.rs:25-44
  std::array<uint8_t, C::SIG_RAW> raw;

  // src/signature.rs:231-253
  static BlsResult<Signature> from_bytes_with_mode(const Bytes& b, SignatureSchemes scheme, SerializationFormat format) {
    auto r = detail::decode<C::SIG_RAW>(C::SIG_GROUP, b.data(), b.size(), C::SIG_BYTES, format);
    if (r.is_err()) return r.unwrap_err();
    return Signature{scheme, r.unwrap()};
  }
  Bytes to_bytes_with_mode(SerializationFormat f) const { return detail::encode(C::SIG_GROUP, raw, C::SIG_BYTES, f); }

  // Vec<u8>::from(&Signature) / Signature::try_from(&[u8]), src/signature.rs:112-126: the serde_bare form -- the enum's variant
  // index as one byte followed by the compressed point, 49 / 97 bytes (the lengths asserted at src/signature.rs:285-286)
  BlsResult<Bytes> to_bytes() const {
    if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
    Bytes out(C::SIG_BYTES + 1);
    const uint8_t tag = (uint8_t)scheme;
    if (int rc = blsgpu_signatures_to_tagged(C::SIG_GROUP, &tag, raw.data(), 1, BLSGPU_FMT_RAW_PROJ, out.data())) return detail::runtime_error(rc);
    return out;
  }
  static BlsResult<Signature> try_from(const Bytes& b) {
    if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
    // serde_bare reports a short buffer or an unknown variant as an error; the reference maps every one to InvalidInputs(..)
    if (b.size() != C::SIG_BYTES + 1) return BlsError{BlsError::Kind::InvalidInputs, "unexpected end of input"};
    Signature s{SignatureSchemes::Basic, {}};
    uint8_t tag = 0;
    int32_t st = 0;
    if (int rc = blsgpu_signatures_from_tagged(C::SIG_GROUP, b.data(), 1, &tag, s.raw.data(), &st)) return detail::runtime_error(rc);
    if (st != BLSGPU_OK) return BlsError{BlsError::Kind::InvalidInputs, tag > 2 ? "invalid variant index" : "invalid point encoding"};
    s.scheme = (SignatureSchemes)tag;
    return s;
  }

  // Signature::verify(&pk, msg), src/signature.rs:130-138: a batch of one
  BlsResult<Unit> verify(const PublicKey<C>& pk, const Bytes& msg) const {
    if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
    const uint64_t offs[2] = {0, msg.size()};
    int32_t st = 0;
    int rc = blsgpu_verify_batch(C::SIG_GROUP, (int)scheme, pk.raw.data(), raw.data(), msg.data(), offs, 1, BLSGPU_FMT_RAW_PROJ, &st);
    if (rc) return detail::runtime_error(rc);
    return detail::from_status(st, nullptr, false);
  }
  // src/signature.rs:177-197 and :256-276
  BlsResult<Unit> verify_secure(const std::vector<PublicKey<C>>& pks, const Bytes& msg) const {
    return verify_secure_with_mode(pks, msg, SerializationFormat::Modern);
  }
  BlsResult<Unit> verify_secure_with_mode(const std::vector<PublicKey<C>>& pks, const Bytes& msg, SerializationFormat format) const {
    if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
    static_assert(sizeof(PublicKey<C>) == C::PK_RAW, "PublicKey must be layout-transparent: the slice crosses the ABI as is");
    int32_t st = 0;
    int rc = blsgpu_verify_secure(C::SIG_GROUP, (int)scheme, pks.data(), pks.size(), raw.data(), msg.data(), msg.size(), (int)format,
                                  BLSGPU_FMT_RAW_PROJ, &st);
    if (rc) return detail::runtime_error(rc);
    return detail::from_status(st, nullptr, false);
  }
};

template <class C>
BlsResult<std::pair<PublicKey<C>, Signature<C>>> SecretKey<C>::sign_with_key(SignatureSchemes scheme, const Bytes& msg) const {
  if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
  const uint64_t offs[2] = {0, msg.size()};
  PublicKey<C> pk;
  Signature<C> sig{scheme, {}};
  int rc = blsgpu_sign_batch(C::SIG_GROUP, (int)scheme, le.data(), msg.data(), offs, 1, pk.raw.data(), sig.raw.data());
  if (rc) return detail::runtime_error(rc);
  return std::make_pair(pk, sig);
}

// NEW, additive: n independent Signature::verify calls in one launch (the reference verifies one signature per call)
template <class C>
std::vector<BlsResult<Unit>> verify_batch(SignatureSchemes scheme, const std::vector<PublicKey<C>>& pks, const std::vector<Bytes>& msgs,
                                          const std::vector<Signature<C>>& sigs) {
  const size_t n = msgs.size();
  std::vector<BlsResult<Unit>> out;
  if (pks.size() != n || sigs.size() != n) return {n, BlsResult<Unit>(BlsError{BlsError::Kind::InvalidInputs, "length mismatch"})};
  if (int rc = detail::ensure_init()) return {n, BlsResult<Unit>(detail::runtime_error(rc))};
  std::vector<uint64_t> offs(n + 1, 0);
  Bytes blob, sraw(n * C::SIG_RAW);
  for (size_t i = 0; i < n; i++) {
    blob.insert(blob.end(), msgs[i].begin(), msgs[i].end());
    offs[i + 1] = blob.size();
    std::memcpy(&sraw[i * C::SIG_RAW], sigs[i].raw.data(), C::SIG_RAW);
  }
  std::vector<int32_t> st(n, 0);
  int rc = blsgpu_verify_batch(C::SIG_GROUP, (int)scheme, pks.data(), sraw.data(), blob.data(), offs.data(), n, BLSGPU_FMT_RAW_PROJ, st.data());
  for (size_t i = 0; i < n; i++) out.push_back(rc ? BlsResult<Unit>(detail::runtime_error(rc)) : detail::from_status(st[i], nullptr, false));
  return out;
}

template <class C>
struct MultiPublicKey {
  PublicKey<C> key;   // the sum, as the reference keeps it (src/multi_public_key.rs:5-11)
  // MultiPublicKey::from_public_keys, src/multi_public_key.rs:79-83 -> BlsMultiKey::from_public_keys (src/traits/pk_multi.rs:7-13)
  static BlsResult<MultiPublicKey> from_public_keys(const std::vector<PublicKey<C>>& pks) {
    auto r = detail::sum<C::PK_RAW>(C::PK_GROUP, (const uint8_t*)pks.data(), pks.size());
    if (r.is_err()) return r.unwrap_err();
    return MultiPublicKey{PublicKey<C>{r.unwrap()}};
  }
};

template <class C>
struct MultiSignature {
  Signature<C> sig;
  // TryFrom<&[Signature<C>]>, src/multi_signature.rs:80-107: fewer than two -> InvalidSignature; a different scheme, or a
  // MessageAugmentation signature after the first, -> InvalidSignatureScheme.  aug_tail_ok: AggregateSignature's variant of
  // the same loop accepts them (src/aggregate_signature.rs:123-146)
  static BlsResult<MultiSignature> from_signatures(const std::vector<Signature<C>>& sigs, bool aug_tail_ok = false) {
    if (sigs.size() < 2) return BlsError{BlsError::Kind::InvalidSignature, ""};
    Bytes raw(sigs.size() * C::SIG_RAW);
    for (size_t i = 0; i < sigs.size(); i++) {
      if (sigs[i].scheme != sigs[0].scheme) return BlsError{BlsError::Kind::InvalidSignatureScheme, ""};
      if (i && !aug_tail_ok && sigs[i].scheme == SignatureSchemes::MessageAugmentation) return BlsError{BlsError::Kind::InvalidSignatureScheme, ""};
      std::memcpy(&raw[i * C::SIG_RAW], sigs[i].raw.data(), C::SIG_RAW);
    }
    auto r = detail::sum<C::SIG_RAW>(C::SIG_GROUP, raw.data(), sigs.size());
    if (r.is_err()) return r.unwrap_err();
    return MultiSignature{Signature<C>{sigs[0].scheme, r.unwrap()}};
  }
  // MultiSignature::verify(mpk, msg), src/multi_signature.rs:127-135
  BlsResult<Unit> verify(const MultiPublicKey<C>& mpk, const Bytes& msg) const { return sig.verify(mpk.key, msg); }
};

template <class C>
struct AggregateSignature {
  Signature<C> sig;
  // AggregateSignature::from_signatures, src/aggregate_signature.rs:166-168 -> TryFrom<&[Signature<C>]> :120-146
  static BlsResult<AggregateSignature> from_signatures(const std::vector<Signature<C>>& sigs) {
    auto r = MultiSignature<C>::from_signatures(sigs, true);
    if (r.is_err()) return r.unwrap_err();
    return AggregateSignature{r.unwrap().sig};
  }
  // AggregateSignature::from_signatures_secure, src/aggregate_signature.rs:191-227 -> aggregate_secure (src/secure_aggregation.rs:110-169)
  static BlsResult<AggregateSignature> from_signatures_secure(const std::vector<Signature<C>>& sigs, const std::vector<PublicKey<C>>& pks,
                                                              SerializationFormat format = SerializationFormat::Modern) {
    if (sigs.size() != pks.size()) return BlsError{BlsError::Kind::InvalidInputs, "Mismatched array lengths"};      // :197-201
    if (sigs.empty()) return BlsError{BlsError::Kind::InvalidInputs, "Empty signatures array"};                     // :203-207
    if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
    Bytes raw(sigs.size() * C::SIG_RAW);
    for (size_t i = 0; i < sigs.size(); i++) {
      if (sigs[i].scheme != sigs[0].scheme) return BlsError{BlsError::Kind::InvalidSignatureScheme, ""};           // :210-212
      std::memcpy(&raw[i * C::SIG_RAW], sigs[i].raw.data(), C::SIG_RAW);
    }
    Signature<C> out{sigs[0].scheme, {}};
    int32_t st = 0;
    int rc = blsgpu_aggregate_secure(C::SIG_GROUP, pks.data(), raw.data(), sigs.size(), (int)format, BLSGPU_FMT_RAW_PROJ, out.raw.data(), &st);
    if (rc) return detail::runtime_error(rc);
    if (st != BLSGPU_OK) return detail::from_status(st, nullptr, false).unwrap_err();
    return AggregateSignature{out};
  }
  // AggregateSignature::verify(&[(pk, msg)]), src/aggregate_signature.rs:230-239
  BlsResult<Unit> verify(const std::vector<std::pair<PublicKey<C>, Bytes>>& data) const {
    if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
    const size_t n = data.size();
    Bytes pks(n * C::PK_RAW), blob;
    std::vector<uint64_t> offs(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
      std::memcpy(&pks[i * C::PK_RAW], data[i].first.raw.data(), C::PK_RAW);
      blob.insert(blob.end(), data[i].second.begin(), data[i].second.end());
      offs[i + 1] = blob.size();
    }
    int32_t st = 0;
    uint64_t aux[2] = {0, 0};
    int rc = blsgpu_aggregate_verify(C::SIG_GROUP, (int)sig.scheme, pks.data(), blob.data(), offs.data(), n, sig.raw.data(), BLSGPU_FMT_RAW_PROJ, &st, aux);
    if (rc) return detail::runtime_error(rc);
    return detail::from_status(st, aux, true);
  }
};

template <class C>
struct ProofOfPossession {
  std::array<uint8_t, C::SIG_RAW> raw;
  // ProofOfPossession::verify(pk), src/proof_of_possession.rs:79-81 -> pop_verify (src/traits/sig_pop.rs:67-70)
  BlsResult<Unit> verify(const PublicKey<C>& pk) const {
    if (int rc = detail::ensure_init()) return detail::runtime_error(rc);
    int32_t st = 0;
    int rc = blsgpu_pop_verify_batch(C::SIG_GROUP, pk.raw.data(), raw.data(), 1, BLSGPU_FMT_RAW_PROJ, &st);
    if (rc) return detail::runtime_error(rc);
    return detail::from_status(st, nullptr, false);
  }
};

// Shamir shares of a key / a signature: identifier + value (vsss_rs share in the reference).  Only the partial-signature
// check is on the accelerated path: PublicKeyShare::verify(sig_share, msg), src/public_key_share.rs:53-72, is the scheme's
// verify on the two share values, i.e. one more item for blsgpu_verify_batch.
template <class C>
struct SignatureShare {
  uint8_t identifier;
  Signature<C> value;
};
template <class C>
struct PublicKeyShare {
  uint8_t identifier;
  PublicKey<C> value;
  BlsResult<Unit> verify(const SignatureShare<C>& sig, const Bytes& msg) const { return sig.value.verify(value, msg); }
};

}  // namespace blsful
