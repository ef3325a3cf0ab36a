// Parity tests written against include/blsful_hip.hpp so that they read like the reference's own tests:
//   tests/cpp_integration_test.rs:87-192      (C++ bls-signatures vectors: sk -> pk, verify, secure aggregation, naive aggregate)
//   tests/secure_aggregation_test.rs:143-235  (57-signer production vector)
//   src/lib.rs / tests of the schemes         (sign / verify round trips, wrong message, multi- and aggregate signatures,
//                                              duplicate-message and identity errors with the reference's strings)
// The vectors come from tests/golden/ref_kats.json (data only), flattened by tests/test_cpp_mirror.py into "key hex" lines.
// Build: g++ -std=c++17 -I include tests/cpp/mirror_test.cpp -L agora-blsful_amd -lblsgpu   (needs a gfx950 device to run)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

#include "blsful_hip.hpp"

using namespace blsful;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                            \
  do {                                                                         \
    g_checks++;                                                                \
    if (!(cond)) {                                                             \
      g_failed++;                                                              \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);            \
    }                                                                          \
  } while (0)

static Bytes unhex(const std::string& h) {
  Bytes b(h.size() / 2);
  for (size_t i = 0; i < b.size(); i++) b[i] = (uint8_t)std::strtoul(h.substr(2 * i, 2).c_str(), nullptr, 16);
  return b;
}
static std::string hex(const Bytes& b) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (uint8_t x : b) {
    s += d[x >> 4];
    s += d[x & 15];
  }
  return s;
}
static std::multimap<std::string, std::string> load(const char* path) {
  std::multimap<std::string, std::string> m;
  std::ifstream f(path);
  std::string line, k, v;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    if (ss >> k >> v) m.emplace(k, v);
  }
  return m;
}
static std::vector<std::string> all(const std::multimap<std::string, std::string>& m, const std::string& k) {
  std::vector<std::string> out;
  auto r = m.equal_range(k);
  for (auto it = r.first; it != r.second; ++it) out.push_back(it->second);
  return out;
}
static Bytes bytes_of(const char* s) { return Bytes(s, s + std::strlen(s)); }

using G2 = Bls12381G2Impl;

// tests/cpp_integration_test.rs:87-192
static void test_cpp_rust_signers(const std::multimap<std::string, std::string>& kat) {
  const Bytes MESSAGE_HELLO = unhex(all(kat, "cpp_message")[0]);
  auto sks = all(kat, "cpp_sk"), pkh = all(kat, "cpp_pk"), sgh = all(kat, "cpp_sig");
  std::vector<PublicKey<G2>> pks;
  std::vector<Signature<G2>> sigs;
  for (size_t i = 0; i < 3; i++) {
    auto sk = SecretKey<G2>::try_from(unhex(sks[i])).unwrap();
    auto pk = PublicKey<G2>::try_from(unhex(pkh[i])).unwrap();
    auto sig = Signature<G2>::from_bytes_with_mode(unhex(sgh[i]), SignatureSchemes::Basic, SerializationFormat::Modern).unwrap();
    CHECK(sk.public_key().unwrap() == pk);                        // "C++ sk should generate pk"
    CHECK(sig.verify(pk, MESSAGE_HELLO).is_ok());                 // verify C++ signatures
    CHECK(sig.verify(pk, bytes_of("hellp")).unwrap_err() == (BlsError{BlsError::Kind::InvalidSignature, ""}));
    pks.push_back(pk);
    sigs.push_back(sig);
  }
  for (size_t n : {2u, 3u}) {                                       // two signers, three signers
    std::vector<PublicKey<G2>> keys(pks.begin(), pks.begin() + n);
    std::vector<Signature<G2>> ss(sigs.begin(), sigs.begin() + n);
    auto secure_agg = AggregateSignature<G2>::from_signatures_secure(ss, keys).unwrap();
    CHECK(secure_agg.sig.scheme == SignatureSchemes::Basic);
    CHECK(secure_agg.sig.verify_secure(keys, MESSAGE_HELLO).is_ok());
    std::vector<PublicKey<G2>> rev(keys.rbegin(), keys.rend());    // the coefficients come from the SORTED keys
    CHECK(secure_agg.sig.verify_secure(rev, MESSAGE_HELLO).is_ok());
    CHECK(secure_agg.sig.verify_secure(keys, bytes_of("other")).is_err());
  }
  // test_normal_aggregation_fails_secure_verify
  auto normal = Signature<G2>::from_bytes_with_mode(unhex(all(kat, "cpp_naive_agg")[0]), SignatureSchemes::Basic, SerializationFormat::Modern).unwrap();
  CHECK(normal.verify_secure({pks[0], pks[1]}, MESSAGE_HELLO).is_err());
  // ... while it is the plain multi-signature of the two
  auto mpk = MultiPublicKey<G2>::from_public_keys({pks[0], pks[1]}).unwrap();
  CHECK((MultiSignature<G2>{normal}).verify(mpk, MESSAGE_HELLO).is_ok());
}

// tests/secure_aggregation_test.rs:143-235
static void test_large_scale_aggregate_signature_verification(const std::multimap<std::string, std::string>& kat) {
  const std::string sig_hex = all(kat, "prod57_sig")[0];
  std::vector<PublicKey<G2>> public_keys;
  for (auto& key_hex : all(kat, "prod57_pk")) {
    auto pk = PublicKey<G2>::try_from(unhex(key_hex));
    CHECK(pk.is_ok());                                            // "Failed to deserialize public key"
    public_keys.push_back(pk.unwrap());
  }
  CHECK(public_keys.size() == 57);
  auto single_sig = Signature<G2>::from_bytes_with_mode(unhex(sig_hex), SignatureSchemes::Basic, SerializationFormat::Modern).unwrap();
  CHECK(sig_hex == hex(single_sig.to_bytes_with_mode(SerializationFormat::Modern)));
  const Bytes message = unhex(all(kat, "prod57_message")[0]);
  CHECK(single_sig.verify_secure(public_keys, message).is_ok());  // "Aggregate signature should verify successfully with secure aggregation"
  CHECK(single_sig.verify_secure_with_mode(public_keys, message, SerializationFormat::Legacy).is_err());   // cross-mode must fail
  public_keys.pop_back();
  CHECK(single_sig.verify_secure(public_keys, message).unwrap_err() == (BlsError{BlsError::Kind::InvalidSignature, ""}));
}

// sign / verify round trips for both implementations and the three schemes, multi- and aggregate signatures, error strings
template <class C>
static void test_schemes(uint8_t tag) {
  std::vector<SecretKey<C>> sks;
  std::vector<PublicKey<C>> pks;
  for (uint8_t i = 0; i < 4; i++) {
    Bytes be(32, 0);
    be[30] = tag;
    be[31] = (uint8_t)(i + 1);
    sks.push_back(SecretKey<C>::try_from(be).unwrap());
    pks.push_back(sks.back().public_key().unwrap());
  }
  const Bytes msg = bytes_of("a message to sign");
  for (auto scheme : {SignatureSchemes::Basic, SignatureSchemes::MessageAugmentation, SignatureSchemes::ProofOfPossession}) {
    auto sig = sks[0].sign(scheme, msg).unwrap();
    CHECK(sig.verify(pks[0], msg).is_ok());
    CHECK(sig.verify(pks[1], msg).unwrap_err().kind == BlsError::Kind::InvalidSignature);
    CHECK(sig.verify(pks[0], bytes_of("another message")).unwrap_err().kind == BlsError::Kind::InvalidSignature);
    // partial signatures: PublicKeyShare::verify(sig_share, msg) (src/public_key_share.rs:53-72)
    PublicKeyShare<C> pk_share{1, pks[0]};
    CHECK(pk_share.verify(SignatureShare<C>{1, sig}, msg).is_ok());
    CHECK((PublicKeyShare<C>{2, pks[1]}).verify(SignatureShare<C>{1, sig}, msg).unwrap_err().kind == BlsError::Kind::InvalidSignature);
    // wire round trip (to_bytes / try_from) keeps the key and the verdict
    auto pk2 = PublicKey<C>::try_from(pks[0].to_bytes()).unwrap();
    CHECK(pk2 == pks[0] && sig.verify(pk2, msg).is_ok());
    // Signature <-> Vec<u8> (serde_bare: scheme tag + compressed point), the reference's try_from test (src/signature.rs:279-318):
    // 49 / 97 bytes, round trip for every scheme, and the decoded signature keeps its scheme and verdict
    {
      const Bytes sb = sig.to_bytes().unwrap();
      CHECK(sb.size() == C::SIG_BYTES + 1 && sb[0] == (uint8_t)scheme);
      auto back = Signature<C>::try_from(sb);
      CHECK(back.is_ok() && back.unwrap().scheme == scheme && back.unwrap().verify(pks[0], msg).is_ok());
      Bytes bad = sb;
      bad[0] = 3;
      CHECK(Signature<C>::try_from(bad).unwrap_err().kind == BlsError::Kind::InvalidInputs);
      CHECK(Signature<C>::try_from(Bytes(sb.begin(), sb.end() - 1)).unwrap_err().kind == BlsError::Kind::InvalidInputs);
    }
    // identity key / identity signature: the reference's check order and strings (src/traits/sig_core.rs:126-135)
    PublicKey<C> inf_pk{};
    Signature<C> inf_sig{scheme, {}};
    CHECK(sig.verify(inf_pk, msg).unwrap_err() == (BlsError{BlsError::Kind::InvalidInputs, "public key is the identity point"}));
    CHECK(inf_sig.verify(inf_pk, msg).unwrap_err() == (BlsError{BlsError::Kind::InvalidInputs, "signature is the identity point"}));
    // the additive batch entry gives the same verdicts as one call per item
    auto res = verify_batch<C>(scheme, {pks[0], pks[1], inf_pk}, {msg, msg, msg}, {sig, sig, sig});
    CHECK(res[0].is_ok() && res[1].unwrap_err().kind == BlsError::Kind::InvalidSignature &&
          res[2].unwrap_err().message == "public key is the identity point");
  }
  // MultiSignature: everyone signs the same message (src/multi_signature.rs, src/multi_public_key.rs)
  {
    std::vector<Signature<C>> sigs;
    for (auto& sk : sks) sigs.push_back(sk.sign(SignatureSchemes::ProofOfPossession, msg).unwrap());
    auto msig = MultiSignature<C>::from_signatures(sigs).unwrap();
    auto mpk = MultiPublicKey<C>::from_public_keys(pks).unwrap();
    CHECK(msig.verify(mpk, msg).is_ok());
    CHECK(msig.verify(MultiPublicKey<C>::from_public_keys({pks[0], pks[1], pks[2]}).unwrap(), msg).is_err());
    CHECK(MultiSignature<C>::from_signatures({sigs[0]}).unwrap_err().kind == BlsError::Kind::InvalidSignature);   // fewer than two
    auto aug = sks[0].sign(SignatureSchemes::MessageAugmentation, msg).unwrap();
    CHECK(MultiSignature<C>::from_signatures({sigs[0], aug}).unwrap_err().kind == BlsError::Kind::InvalidSignatureScheme);
  }
  // AggregateSignature: distinct messages; duplicates are rejected under Basic before any pairing (src/traits/sig_basic.rs:41-64)
  for (auto scheme : {SignatureSchemes::Basic, SignatureSchemes::MessageAugmentation, SignatureSchemes::ProofOfPossession}) {
    std::vector<Signature<C>> sigs;
    std::vector<std::pair<PublicKey<C>, Bytes>> data;
    for (size_t i = 0; i < sks.size(); i++) {
      Bytes m = bytes_of("message number ");
      m.push_back((uint8_t)('0' + i));
      sigs.push_back(sks[i].sign(scheme, m).unwrap());
      data.push_back({pks[i], m});
    }
    auto asig = AggregateSignature<C>::from_signatures(sigs).unwrap();
    CHECK(asig.verify(data).is_ok());
    auto bad = data;
    bad[2].second = bytes_of("tampered");
    CHECK(asig.verify(bad).unwrap_err().kind == BlsError::Kind::InvalidSignature);
    auto dup = data;
    dup[3].second = dup[1].second;
    auto r = asig.verify(dup);
    if (scheme == SignatureSchemes::Basic)
      CHECK(r.unwrap_err() == (BlsError{BlsError::Kind::InvalidInputs, "duplicate messages detected at 1 and 3"}));
    else
      CHECK(r.unwrap_err().kind == BlsError::Kind::InvalidSignature);
    auto idk = data;
    idk[1].first = PublicKey<C>{};
    CHECK(asig.verify(idk).unwrap_err() == (BlsError{BlsError::Kind::InvalidInputs, "public key at 2 is the identity point"}));
  }
  // secure aggregation round trip, both serialisation modes where they exist (Legacy: 48-byte keys only)
  {
    std::vector<Signature<C>> sigs;
    for (auto& sk : sks) sigs.push_back(sk.sign(SignatureSchemes::Basic, msg).unwrap());
    auto agg = AggregateSignature<C>::from_signatures_secure(sigs, pks).unwrap();
    CHECK(agg.sig.verify_secure(pks, msg).is_ok());
    CHECK(agg.sig.verify_secure({pks[0], pks[1], pks[2]}, msg).is_err());
    CHECK(AggregateSignature<C>::from_signatures_secure(sigs, {pks[0]}).unwrap_err() ==
          (BlsError{BlsError::Kind::InvalidInputs, "Mismatched array lengths"}));
    CHECK(AggregateSignature<C>::from_signatures_secure({}, {}).unwrap_err() == (BlsError{BlsError::Kind::InvalidInputs, "Empty signatures array"}));
    if (C::PK_BYTES == 48) {
      auto lagg = AggregateSignature<C>::from_signatures_secure(sigs, pks, SerializationFormat::Legacy).unwrap();
      CHECK(lagg.sig.verify_secure_with_mode(pks, msg, SerializationFormat::Legacy).is_ok());
      CHECK(lagg.sig.verify_secure_with_mode(pks, msg, SerializationFormat::Modern).is_err());
      auto rt = PublicKey<C>::from_bytes_with_mode(pks[0].to_bytes_with_mode(SerializationFormat::Legacy), SerializationFormat::Legacy);
      CHECK(rt.is_ok() && rt.unwrap() == pks[0]);
    }
  }
  // wire decoding errors (src/public_key.rs:159-164, src/impls/legacy.rs:71-82)
  CHECK(PublicKey<C>::try_from(Bytes(C::PK_BYTES - 1, 0)).unwrap_err().kind == BlsError::Kind::InvalidLength);
  CHECK(PublicKey<C>::try_from(Bytes(C::PK_BYTES, 0)).is_err());
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("usage: %s <kat file>\n", argv[0]);
    return 2;
  }
  // no device: every call must fail loudly (there is no CPU path behind the mirror)
  {
    Bytes be(32, 0);
    be[31] = 1;
    auto probe = SecretKey<G2>::try_from(be).unwrap().public_key();
    if (probe.is_err()) {
      std::printf("NO DEVICE: %s\n", probe.unwrap_err().message.c_str());
      return probe.unwrap_err().kind == BlsError::Kind::Runtime ? 3 : 1;
    }
  }
  auto kat = load(argv[1]);
  test_cpp_rust_signers(kat);
  test_large_scale_aggregate_signature_verification(kat);
  test_schemes<Bls12381G1Impl>(0x11);
  test_schemes<Bls12381G2Impl>(0x22);
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
