"""Pins of the oracle itself (CPU only): the reference's own known-answer vectors, RFC 9380 vectors, algebraic
identities, and the reference's legacy-format tests restated."""
import hashlib
import random

import pytest

import util
from util import c, ref
from oracle.py import check_kats, iso_consts


def test_reference_kats():
    """K1-K4 of SURVEY 8c: tests/cpp_integration_test.rs:19-82,87-192 and tests/secure_aggregation_test.rs:143-235."""
    assert check_kats.run(verbose=False)


def test_rfc9380_vectors():
    """RFC 9380 J.9.1 (G1) and J.10.1 (G2) hash_to_curve vectors for msg = "" / "abc" (published spec, not reference)."""
    d1 = b'QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_'
    p = c.hash_to_g1(b'', d1)
    assert p == (0x052926add2207b76ca4fa57a8734416c8dc95e24501772c814278700eed6d1e4e8cf62d9c09db0fac349612b759e79a1,
                 0x08ba738453bfed09cb546dbb0783dbb3a5f1f566ed67bb6be0e8c67e2e81a4cc68ee29813bb7994998f3eae0c9c6a265)
    p = c.hash_to_g1(b'abc', d1)
    assert p == (0x03567bc5ef9c690c2ab2ecdf6a96ef1c139cc0b2f284dca0a9a7943388a49a3aee664ba5379a7655d3c68900be2f6903,
                 0x0b9c15f3fe6e5cf4211f346271d7b01c8f3b28be689c8429c85b67af215533311f0b8dfaaa154fa6b88176c229f2885d)
    d2 = b'QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_'
    q = c.hash_to_g2(b'', d2)
    assert q[0] == (0x0141ebfbdca40eb85b87142e130ab689c673cf60f1a3e98d69335266f30d9b8d4ac44c1038e9dcdd5393faf5c41fb78a,
                    0x05cb8437535e20ecffaef7752baddf98034139c38452458baeefab379ba13dff5bf5dd71b72418717047f5b0f37da03d)


def test_rfc9380_oversize_dst_vector():
    """RFC 9380 section 5.3.3 / appendix K.2: a DST longer than 255 bytes enters expand_message_xmd as
    SHA-256("H2C-OVERSIZE-DST-" || DST).  The K.2 vector (256-byte DST, msg = "", 32 bytes out) pins the rule the boundary
    inherits from the reference's dependency (reference src/traits/hash_to_point.rs:11 takes any DST)."""
    dst = b'QUUX-V01-CS02-with-expander-SHA256-128-long-DST-' + b'1' * 208
    assert len(dst) == 256
    assert hashlib.sha256(b'H2C-OVERSIZE-DST-' + dst).hexdigest() == '412717974da474d0f8c420f320ff81e8432adb7c927d9bd082b4fb4d16c0a236'
    assert c.expand_message_xmd(b'', dst, 0x20).hex() == 'e8dc0c8b686b7ef2074086fbdd2f30e3f8bfbd3bdf177f73f04b97ce618a3ed3'
    # at 255 bytes the DST is used as it is
    d255 = dst[:255]
    b0 = hashlib.sha256(bytes(64) + b'm' + (32).to_bytes(2, 'big') + b'\x00' + d255 + b'\xff').digest()
    assert c.expand_message_xmd(b'm', d255, 32) == hashlib.sha256(b0 + b'\x01' + d255 + b'\xff').digest()


def test_sswu_curves_and_isogenies():
    """The derived isogeny tables: E' has the order of E, the maps land on E and are group homomorphisms."""
    rng = random.Random(5)
    # #E'1(Fp) = #E1(Fp) = h1 * r
    while True:
        x = rng.randrange(c.P)
        y = c.fp_sqrt(c.E1_ISO.rhs(x))
        if y is not None:
            break
    assert c.E1_ISO.mul((x, y), c.H1 * c.R) is None
    for _ in range(3):
        u, v = rng.randrange(c.P), rng.randrange(c.P)
        a, b = c._sswu(c.E1_ISO, iso_consts.G1_Z, u, c.fp_is_square, c.fp_sqrt, c._sgn0_fp, 1), \
            c._sswu(c.E1_ISO, iso_consts.G1_Z, v, c.fp_is_square, c.fp_sqrt, c._sgn0_fp, 1)
        assert c.E1_ISO.on_curve(a) and c.E1_ISO.on_curve(b)
        ia, ib, iab = (c._iso_map(c.E1_ISO, iso_consts.G1_ISO, t) for t in (a, b, c.E1_ISO.add(a, b)))
        assert c.E1.on_curve(ia) and c.E1.add(ia, ib) == iab
        u2, v2 = (rng.randrange(c.P), rng.randrange(c.P)), (rng.randrange(c.P), rng.randrange(c.P))
        a, b = c._sswu(c.E2_ISO, iso_consts.G2_Z, u2, c.f2_is_square, c.f2_sqrt, c._sgn0_f2, c.F2_ONE), \
            c._sswu(c.E2_ISO, iso_consts.G2_Z, v2, c.f2_is_square, c.f2_sqrt, c._sgn0_f2, c.F2_ONE)
        ia, ib, iab = (c._iso_map(c.E2_ISO, iso_consts.G2_ISO, t) for t in (a, b, c.E2_ISO.add(a, b)))
        assert c.E2.on_curve(ia) and c.E2.add(ia, ib) == iab
    assert c.g1_in_subgroup(c.hash_to_g1(b'x', b'dst')) and c.g2_in_subgroup(c.hash_to_g2(b'x', b'dst'))
    assert c.g1_in_subgroup(c.G1_GEN) and c.g2_in_subgroup(c.G2_GEN)


def test_pairing_identities():
    rng = random.Random(6)
    a, b = rng.randrange(1, c.R), rng.randrange(1, c.R)
    P, Q = c.E1.mul(c.G1_GEN, a), c.E2.mul(c.G2_GEN, b)
    e1 = c.final_exponentiation(c.miller_loop([(P, Q)]))
    e2 = c.f12_pow(c.final_exponentiation(c.miller_loop([(c.G1_GEN, c.G2_GEN)])), a * b % c.R)
    assert e1 == e2 and e1 != c.F12_ONE
    # the fast final exponentiation is the cube of the canonical one
    f = c.miller_loop([(P, Q)])
    assert c.final_exponentiation(f) == c.f12_pow(c.final_exponentiation_naive(f), 3)
    assert c.pairing_product_is_one([(P, Q), (c.E1.neg(c.E1.mul(c.G1_GEN, a * b % c.R)), c.G2_GEN)])
    assert c.pairing_product_is_one([(None, Q), (P, None)])


def test_secure_coefficients_golden():
    """SHA-256-only values derived from the reference's C++ keys (tests/cpp_integration_test.rs:35-51)."""
    import json, os
    k = json.load(open(os.path.join(util.ROOT, 'tests', 'golden', 'ref_kats.json')))
    pk = [bytes.fromhex(h) for h in k['cpp']['pk']]
    perm, H, ts = ref.secure_coefficients(pk[:2])
    assert perm == [1, 0] and H.hex() == '6040b788e954eb9df1a0d581cf020f7b1946d0ed0dd48de1d03bab45e3b3a29a'
    assert ts[0] == 0x584ccd89aaf51f8b06067b165b36a9096ae4abc23189c97ca1d34accb015244a
    perm, H, ts = ref.secure_coefficients([ref.modern_to_legacy(b) for b in pk])
    assert perm == [2, 1, 0] and H.hex() == '88ec5f152a807e8f64a0639972defc88caa305a548fe290897e27ef2a052f140'
    assert ts[2] == 0x48b0c8fe31dde82cb08bb7666233296901c34e121f19653d3b238ade1a5d971c


def test_legacy_format():
    """reference src/impls/legacy.rs:172-253, tests/legacy_test.rs:33-34, tests/legacy_comprehensive_test.rs:212-288."""
    rng = random.Random(8)
    C = ref.G2Impl
    assert ref.modern_to_legacy(c.g1_compress(None)) == c.g1_compress(None) and c.g1_compress(None)[0] == 0xc0
    for _ in range(100):
        pt = c.E1.mul(c.G1_GEN, rng.randrange(1, c.R))
        mod = c.g1_compress(pt)
        leg = ref.modern_to_legacy(mod)
        assert leg[0] & 0x60 == 0 and (leg[0] >> 7) == (mod[0] >> 5 & 1) and leg[1:] == mod[1:]
        assert ref.legacy_to_modern(leg) == mod
        assert ref.pk_from_bytes_with_mode(C, leg, ref.LEGACY) == pt == ref.pk_from_bytes_with_mode(C, mod, ref.MODERN)
    with pytest.raises(ref.BlsError) as e:
        ref.legacy_to_modern(bytes([0x20]) + bytes(47))
    assert e.value.kind == 'LegacyFormatError'
    with pytest.raises(ref.BlsError) as e:
        ref.pk_from_bytes_with_mode(C, bytes([0x40]) + bytes(47), ref.MODERN)
    assert e.value.kind == 'DeserializationError'
    with pytest.raises(ref.BlsError) as e:
        ref.pk_from_bytes_with_mode(C, bytes(47), ref.MODERN)
    assert e.value.kind == 'InvalidLength'
    sig = c.E2.mul(c.G2_GEN, 77)
    b = c.g2_compress(sig)
    assert ref.sig_from_bytes_with_mode(C, ref.modern_to_legacy(b), ref.LEGACY) == sig


def test_scheme_semantics():
    """reference tests/signatures.rs:133-173 on the oracle: Basic rejects duplicate messages, Aug accepts them."""
    C = ref.G2Impl
    sks = [ref.keygen_from_hash(bytes([i]) * 32) for i in range(3)]
    pks = [ref.public_key(C, s) for s in sks]
    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        sig = ref.aggregate_signatures(C, [ref.sign(C, scheme, s, b'same') for s in sks])
        if scheme == ref.BASIC:
            with pytest.raises(ref.BlsError) as e:
                ref.aggregate_verify(C, scheme, [(p, b'same') for p in pks], sig)
            assert e.value == ref.InvalidInputs('duplicate messages detected at 0 and 1')
        else:
            ref.aggregate_verify(C, scheme, [(p, b'same') for p in pks], sig)
    with pytest.raises(ref.BlsError) as e:
        ref.aggregate_verify(C, ref.POP, [(pks[0], b'a'), (None, b'b')], sig)
    assert e.value == ref.InvalidInputs('public key at 2 is the identity point')
