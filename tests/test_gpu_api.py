"""GPU parity for the rest of the C ABI: MultiSignature::verify, AggregateSignature::verify, verify_secure[_with_mode],
hash_to_point, point sums / MSM, pairing products, serialisation.  Each test cites the reference test it mirrors."""
import ctypes
import hashlib
import json
import os
import random

import pytest

import util
from util import c, ref

pytestmark = pytest.mark.gpu

IMPLS = [(ref.G1Impl, 1), (ref.G2Impl, 2)]
KATS = json.load(open(os.path.join(util.ROOT, 'tests', 'golden', 'ref_kats.json')))


def raw_fns(sg):
    return (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)


def keys(C, n, tag=0):
    sks = [ref.keygen_from_hash(bytes([tag]) + i.to_bytes(4, 'big') + bytes(27)) for i in range(n)]
    return sks, [ref.public_key(C, s) for s in sks]


# ------------------------------------------------------------------ hash_to_point / group ops
@pytest.mark.parametrize('group', [1, 2])
def test_hash_to_point(api, group):
    """HashToPoint::hash_to_point (reference src/impls/g1.rs:17-19, g2.rs:15-17) incl. RFC 9380 J.9.1/J.10.1 inputs."""
    msgs = [b'', b'abc', b'abcdef0123456789', b'q128_' + b'q' * 128, b'a512_' + b'a' * 512, bytes(range(256))]
    dst = (b'QUUX-V01-CS02-with-BLS12381G%d_XMD:SHA-256_SSWU_RO_' % group)
    got = api.hash_to_point(group, msgs, dst)
    comp = api.serialize(group, got)
    for m, b in zip(msgs, comp):
        want = c.g1_compress(c.hash_to_g1(m, dst)) if group == 1 else c.g2_compress(c.hash_to_g2(m, dst))
        assert b == want
    assert api.hash_to_point(group, [], dst) == []


@pytest.mark.gpu
@pytest.mark.parametrize('group', [1, 2])
def test_hash_to_point_and_core_verify_take_any_dst(api, group):
    """DSTs of 255 / 256 / 300 bytes through blsgpu_hash_to_g1/g2 and blsgpu_core_verify against the oracle (VERDICT r3 missing #3):
    the interface takes any DST (reference src/traits/hash_to_point.rs:11); from 256 bytes on expand_message_xmd uses
    SHA-256("H2C-OVERSIZE-DST-" || DST) (RFC 9380 5.3.3, pinned by the K.2 vector in tests/test_oracle.py)."""
    rng = random.Random(77 + group)
    msgs = [b'', b'abc', bytes(range(200))]
    C = ref.G1Impl if group == 1 else ref.G2Impl
    sk = ref.keygen_from_hash(b'\x55' * 32)
    pk = ref.public_key(C, sk)
    pk_raw = util.g2_raw(pk, rng) if group == 1 else util.g1_raw(pk, rng)
    for dl in (255, 256, 300, 1000):
        dst = ((b'oversize-dst-%d-' % group) * 80)[:dl]
        comp = api.serialize(group, api.hash_to_point(group, msgs, dst))
        for m, b in zip(msgs, comp):
            want = c.g1_compress(c.hash_to_g1(m, dst)) if group == 1 else c.g2_compress(c.hash_to_g2(m, dst))
            assert b == want, (dl, m)
        # a signature under that DST verifies through core_verify; under the neighbouring length it does not
        H = C.hash_to_point(b'abc', dst)
        sig = C.sig_curve.mul(H, sk)
        sig_raw = util.g1_raw(sig, rng) if group == 1 else util.g2_raw(sig, rng)
        assert api.core_verify(group, dst, [pk_raw], [sig_raw], [b'abc']) == [0]
        assert api.core_verify(group, dst + b'x', [pk_raw], [sig_raw], [b'abc']) == [1]


@pytest.mark.parametrize('group', [1, 2])
def test_sum_and_msm(api, group):
    """aggregate_public_keys (reference src/traits/sig_core.rs:50-59) and sum t_i pk_i (src/secure_aggregation.rs:201-204)."""
    rng = random.Random(group)
    E, gen, raw, comp = (c.E1, c.G1_GEN, util.g1_raw, c.g1_compress) if group == 1 else (c.E2, c.G2_GEN, util.g2_raw, c.g2_compress)
    for n in (1, 2, 3, 63, 64, 65, 300):
        ks = [rng.randrange(1, c.R) for _ in range(n)]
        pts = [E.mul(gen, k) for k in ks]
        if n >= 3:
            pts[1] = None                       # an identity among the inputs
            pts[2] = pts[0]                     # a duplicate (forces the doubling branch of the adder)
        want = None
        for p in pts:
            want = E.add(want, p)
        got = api.serialize(group, [api.point_sum(group, [raw(p, rng) for p in pts])])[0]
        assert got == comp(want), n
        if n <= 65:
            scal = [rng.randrange(c.R) for _ in range(n)]
            if n >= 2:
                scal[0] = 0
            want = None
            for p, s in zip(pts, scal):
                want = E.add(want, E.mul(p, s))
            got = api.serialize(group, [api.point_sum(group, [raw(p, rng) for p in pts], scal)])[0]
            assert got == comp(want), n
    assert api.serialize(group, [api.point_sum(group, [])])[0] == comp(None)


def test_sum_closed_form_large(api):
    """Size-independent property at a large size: sum_i (s0 + i) g2 == (n s0 + n(n-1)/2) g2 for n = 20,000 device-made keys."""
    n, s0 = 20000, 0x1234567
    pks, _ = api.sign_batch(1, api.POP, [s0 + i for i in range(n)], [b'x'] * n)
    got = api.serialize(2, [api.point_sum(2, pks)])[0]
    assert got == c.g2_compress(c.E2.mul(c.G2_GEN, (n * s0 + n * (n - 1) // 2) % c.R))


@pytest.mark.parametrize('group', [1, 2])
def test_serialize_modes(api, group):
    """to_bytes / to_bytes_with_mode incl. infinity (reference src/impls/legacy.rs:19-35,85-98,129-143; :204)."""
    rng = random.Random(9)
    E, gen, raw, comp = (c.E1, c.G1_GEN, util.g1_raw, c.g1_compress) if group == 1 else (c.E2, c.G2_GEN, util.g2_raw, c.g2_compress)
    pts = [E.mul(gen, rng.randrange(1, c.R)) for _ in range(20)] + [None]
    raws = [raw(p, rng) for p in pts]
    assert api.serialize(group, raws) == [comp(p) for p in pts]
    assert api.serialize(group, raws, legacy=True) == [ref.modern_to_legacy(comp(p)) for p in pts]
    aff = [(util.g1_aff_raw(p) if group == 1 else util.g2_aff_raw(p)) if p else bytes(96 * group) for p in pts]
    assert api.serialize(group, aff, fmt_in=api.FMT_RAW_AFFINE) == [comp(p) for p in pts]


def test_pairing_product(api):
    """Pairing::pairing(..).is_identity() (reference src/traits/pairings.rs:50, src/helpers.rs:41-63): bilinearity."""
    rng = random.Random(4)
    a, b = rng.randrange(1, c.R), rng.randrange(1, c.R)
    P, Q = c.E1.mul(c.G1_GEN, a), c.E2.mul(c.G2_GEN, b)
    nP = c.E1.neg(c.E1.mul(c.G1_GEN, a * b % c.R))
    assert api.pairing_product_is_one([util.g1_raw(P, rng), util.g1_raw(nP, rng)], [util.g2_raw(Q, rng), util.g2_raw(c.G2_GEN, rng)])
    assert not api.pairing_product_is_one([util.g1_raw(P, rng), util.g1_raw(nP, rng)], [util.g2_raw(Q, rng), util.g2_raw(Q, rng)])
    assert api.pairing_product_is_one([util.g1_raw(None), util.g1_raw(P, rng)], [util.g2_raw(Q, rng), util.g2_raw(None)])
    assert api.pairing_product_is_one([], [])
    # 5 pairs: e(k_i G1, G2) for sum k_i = 0
    ks = [rng.randrange(c.R) for _ in range(4)]
    ks.append(-sum(ks) % c.R)
    assert api.pairing_product_is_one([util.g1_raw(c.E1.mul(c.G1_GEN, k), rng) for k in ks], [util.g2_raw(c.G2_GEN, rng)] * 5)
    # 70 and 300 pairs (from 64 pairs on the product is taken entry by entry over the items, blsgpu.hip run_miller_product_tree):
    # e(k_i G1, m_i G2) with sum k_i m_i = 0, identities sprinkled in on either side; one k off by one must fail
    for n in (70, 300):
        ms = [rng.randrange(1, 50) for _ in range(n)]
        ks = [rng.randrange(c.R) for _ in range(n - 1)]
        ks.append(-sum(k * m for k, m in zip(ks, ms)) * pow(ms[-1], -1, c.R) % c.R)
        g2s = {m: c.E2.mul(c.G2_GEN, m) for m in set(ms)}
        p1 = [util.g1_raw(c.E1.mul(c.G1_GEN, k), rng) for k in ks] + [util.g1_raw(None), util.g1_raw(c.G1_GEN, rng)]
        p2 = [util.g2_raw(g2s[m], rng) for m in ms] + [util.g2_raw(c.G2_GEN, rng), util.g2_raw(None)]
        assert api.pairing_product_is_one(p1, p2), n
        p1[n // 2] = util.g1_raw(c.E1.mul(c.G1_GEN, (ks[n // 2] + 1) % c.R), rng)
        assert not api.pairing_product_is_one(p1, p2), n


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_sign_batch_matches_oracle(api, C, sg):
    """Sign side used for synthetic inputs: core_sign (reference src/traits/sig_core.rs:108-117) incl. Aug prefix."""
    sks, pks = keys(C, 5, 3)
    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        msgs = [b'm%d' % i for i in range(5)]
        dpks, dsigs = api.sign_batch(sg, scheme, sks, msgs)
        assert api.serialize(3 - sg, dpks) == [C.pk_to_bytes(p) for p in pks]
        assert api.serialize(sg, dsigs) == [C.sig_to_bytes(ref.sign(C, scheme, s, m)) for s, m in zip(sks, msgs)]


# ------------------------------------------------------------------ MultiSignature::verify
@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_multisig(api, pkg, C, sg):
    """reference tests/signatures.rs:91-128: all signers ok, a missing key fails; plus identity aggregate key."""
    rng = random.Random(sg)
    pkraw, sigraw = raw_fns(sg)
    impl = pkg.Bls12381G1Impl if sg == 1 else pkg.Bls12381G2Impl
    sks, pks = keys(C, 6, 5)
    msg = b'signatures_work'
    for scheme in (ref.BASIC, ref.POP, ref.AUG):
        if scheme == ref.AUG:   # the aggregate key is the augmentation prefix (reference src/traits/sig_aug.rs:20-24)
            apk = ref.aggregate_public_keys(C, pks)
            sig = None
            for s in sks:
                sig = C.sig_curve.add(sig, C.sig_curve.mul(C.hash_to_point(C.pk_to_bytes(apk) + msg, C.DST[scheme]), s))
        else:
            sig = ref.aggregate_signatures(C, [ref.sign(C, scheme, s, msg) for s in sks])
        msig = pkg.MultiSignature(impl, scheme, sigraw(sig, rng))
        mpk = pkg.MultiPublicKey.from_public_keys([pkg.PublicKey(impl, pkraw(p, rng)) for p in pks])
        msig.verify(mpk, msg)
        with pytest.raises(pkg.BlsError) as e:
            msig.verify(pkg.MultiPublicKey.from_public_keys([pkg.PublicKey(impl, pkraw(p, rng)) for p in pks[1:]]), msg)
        assert e.value == pkg.BlsError('InvalidSignature')
        for bad in ([pks[0], C.pk_curve.neg(pks[0])], []):
            try:
                ref.multi_sig_verify(C, scheme, bad, sig, msg)
                want = None
            except ref.BlsError as ex:
                want = (ex.kind, ex.msg)
            with pytest.raises(pkg.BlsError) as e:
                pkg.MultiSignature(impl, scheme, sigraw(sig, rng)).verify(pkg.MultiPublicKey(impl, [pkg.PublicKey(impl, pkraw(p, rng)) for p in bad]), msg)
            assert (e.value.kind, e.value.msg) == want


# ------------------------------------------------------------------ AggregateSignature::verify
@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_aggregate_verify(api, pkg, C, sg):
    """reference tests/signatures.rs:133-173 (same message under Basic must fail, distinct ok, same message under Aug ok)
    and the error order of src/traits/sig_basic.rs:41-64 / src/traits/sig_core.rs:149-178."""
    rng = random.Random(20 + sg)
    pkraw, sigraw = raw_fns(sg)
    impl = pkg.Bls12381G1Impl if sg == 1 else pkg.Bls12381G2Impl
    n = 9
    sks, pks = keys(C, n, 9)

    def both(scheme, items, sig):
        try:
            ref.aggregate_verify(C, scheme, items, sig)
            want = None
        except ref.BlsError as ex:
            want = (ex.kind, ex.msg)
        agg = pkg.AggregateSignature(impl, scheme, sigraw(sig, rng))
        data = [(pkg.PublicKey(impl, pkraw(p, rng)), m) for p, m in items]
        try:
            agg.verify(data)
            got = None
        except pkg.BlsError as ex:
            got = (ex.kind, ex.msg)
        assert got == want, (scheme, got, want)
        return got

    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        msgs = [b'message %d' % i + bytes(i) for i in range(n)]
        sig = ref.aggregate_signatures(C, [ref.sign(C, scheme, s, m) for s, m in zip(sks, msgs)])
        assert both(scheme, list(zip(pks, msgs)), sig) is None
        assert both(scheme, list(zip(pks, msgs))[:-1], sig) == ('InvalidSignature', '')
        same = [b'same'] * n
        sig2 = ref.aggregate_signatures(C, [ref.sign(C, scheme, s, m) for s, m in zip(sks, same)])
        r = both(scheme, list(zip(pks, same)), sig2)
        assert (r is not None) == (scheme == ref.BASIC)
        # error precedence: duplicates (Basic) > signature identity > first identity key (1-based) > pairing
        pk_id = list(pks)
        pk_id[4] = None
        pk_id[6] = None
        both(scheme, list(zip(pk_id, msgs)), sig)
        both(scheme, list(zip(pk_id, msgs)), None)
        dup = list(msgs)
        dup[7] = dup[2]
        both(scheme, list(zip(pk_id, dup)), None)
        both(scheme, [], sig)
        both(scheme, [], None)


def test_aggregate_verify_many(api):
    """A larger aggregate (n = 600, product tree over several folds) built on the device; one tampered message flips it."""
    n = 600
    sks = [0x1000 + 7 * i for i in range(n)]
    msgs = [hashlib.sha256(b'agg%d' % i).digest() for i in range(n)]
    pks, sigs = api.sign_batch(1, api.BASIC, sks, msgs)
    agg = api.point_sum(1, sigs)
    assert api.aggregate_verify(1, api.BASIC, pks, msgs, agg) == (api.OK, (0, 0))
    msgs2 = list(msgs)
    msgs2[417] = b'tampered'
    assert api.aggregate_verify(1, api.BASIC, pks, msgs2, agg)[0] == api.INVALID_SIGNATURE
    msgs2[599] = msgs2[3]
    assert api.aggregate_verify(1, api.BASIC, pks, msgs2, agg) == (api.DUPLICATE_MESSAGE, (3, 599))
    assert api.aggregate_verify(1, api.POP, pks, msgs2, agg)[0] == api.INVALID_SIGNATURE


# ------------------------------------------------------------------ verify_secure
def test_secure_coefficients_golden(api):
    """SHA-256-only golden values (inputs = reference tests/cpp_integration_test.rs:35-51)."""
    pk = [bytes.fromhex(h) for h in KATS['cpp']['pk']]
    st, perm, ts = api.secure_coefficients(pk[:2])
    assert (st, perm) == (0, [1, 0])
    assert ts == [0x584ccd89aaf51f8b06067b165b36a9096ae4abc23189c97ca1d34accb015244a,
                  0x350f133013a3e8f028ab28c14c710b88cc15bfaa853497887728fe591c20a17d]
    st, perm, ts = api.secure_coefficients(pk)
    assert (st, perm) == (0, [2, 1, 0])
    assert ts == [0x06affd8cb2dc37f9c3c4b15a8e7dc6c9b12a845877d24eaa2e2be6628dda7755,
                  0x5bcd568774ca9fbe351d45b43e504a2f17d6d169142a4286414e90a9de5b01a8,
                  0x07a4139aa0177dbc814d996431d547dca201ca30ff948d57c7d40bfed02f3c9e]
    leg = [ref.modern_to_legacy(b) for b in pk]
    st, perm, ts = api.secure_coefficients(leg)
    assert (st, perm) == (0, [2, 1, 0])
    assert ts[0] == 0x43cc66d4a23309b3e0d7c75dc3d2d04df70e6f9f6d826b025ecaf81371906b00
    # against the oracle on random keys, with duplicates (stable order)
    rng = random.Random(1)
    kb = [bytes(rng.randrange(256) for _ in range(96)) for _ in range(40)]
    kb[7] = kb[30]
    perm_o, _, ts_o = ref.secure_coefficients(kb)
    assert api.secure_coefficients(kb) == (0, perm_o, ts_o)
    # shared 8-byte prefixes (the device library radix-sorts the prefix word and comparison-sorts the runs), keys that
    # differ only in the last byte, exact duplicates far apart: the order must stay the stable byte-lexicographic one
    for width in (48, 96):
        kb = []
        for i in range(300):
            prefix = bytes([rng.randrange(3)] * 8)
            body = bytes(rng.randrange(2) for _ in range(width - 9)) + bytes([rng.randrange(256)])
            kb.append(prefix + body)
        kb[250] = kb[3]
        kb[299] = kb[3]
        perm_o, _, ts_o = ref.secure_coefficients(kb)
        assert api.secure_coefficients(kb) == (0, perm_o, ts_o)


def test_verify_secure_reference_kats(api, pkg):
    """K2-K4 through the C ABI: reference tests/cpp_integration_test.rs:87-192 and
    tests/secure_aggregation_test.rs:143-235 (57-signer production vector)."""
    C, impl = ref.G2Impl, pkg.Bls12381G2Impl
    cpp = KATS['cpp']
    msg = bytes.fromhex(cpp['message'])
    pks = [C.pk_from_bytes(bytes.fromhex(h)) for h in cpp['pk']]
    sigs = [C.sig_from_bytes(bytes.fromhex(h)) for h in cpp['sig']]
    P = [pkg.PublicKey(impl, util.g1_raw(p)) for p in pks]
    for n in (2, 3):
        agg = ref.aggregate_secure(C, pks[:n], sigs[:n])
        pkg.Signature(impl, pkg.BASIC, util.g2_raw(agg)).verify_secure(P[:n], msg)
        pkg.Signature(impl, pkg.BASIC, util.g2_raw(agg)).verify_secure(P[:n][::-1], msg)     # order independent
    naive = C.sig_from_bytes(bytes.fromhex(cpp['naive_agg_sig_pk12']))
    with pytest.raises(pkg.BlsError) as e:
        pkg.Signature(impl, pkg.BASIC, util.g2_raw(naive)).verify_secure(P[:2], msg)
    assert e.value == pkg.BlsError('InvalidSignature')
    p57 = KATS['prod57']
    P57 = [pkg.PublicKey(impl, util.g1_raw(C.pk_from_bytes(bytes.fromhex(h)))) for h in p57['pks']]
    sig57 = pkg.Signature(impl, pkg.BASIC, util.g2_raw(C.sig_from_bytes(bytes.fromhex(p57['sig']))))
    sig57.verify_secure(P57, bytes.fromhex(p57['message']))
    sig57.verify_secure_with_mode(P57, bytes.fromhex(p57['message']), pkg.MODERN)
    with pytest.raises(pkg.BlsError):
        sig57.verify_secure_with_mode(P57, bytes.fromhex(p57['message']), pkg.LEGACY)       # cross-mode must fail
    with pytest.raises(pkg.BlsError):
        sig57.verify_secure(P57[:-1], bytes.fromhex(p57['message']))


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_verify_secure_selfconsistency(api, pkg, C, sg):
    """reference src/secure_aggregation.rs:433-602 (3 signers ok / wrong subset / reorder; rogue key; empty keys) and
    tests/legacy_test.rs:110-171 (cross-mode verification fails)."""
    rng = random.Random(sg)
    pkraw, sigraw = raw_fns(sg)
    impl = pkg.Bls12381G1Impl if sg == 1 else pkg.Bls12381G2Impl
    sks, pks = keys(C, 5, 11)
    msg = b'test message'
    modes = [None] if sg == 1 else [None, ref.MODERN, ref.LEGACY]
    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        sigs = [C.sig_curve.mul(C.hash_to_point(msg, C.DST[scheme]), s) for s in sks]   # Aug does not prefix here (:236-246)
        for mode in modes:
            agg = ref.aggregate_secure(C, pks, sigs, mode)
            S = pkg.Signature(impl, scheme, sigraw(agg, rng))
            P = [pkg.PublicKey(impl, pkraw(p, rng)) for p in pks]

            def run(keys_, m=msg, sig=S):
                try:
                    if mode is None:
                        sig.verify_secure(keys_, m)
                    else:
                        sig.verify_secure_with_mode(keys_, m, mode)
                    return None
                except pkg.BlsError as ex:
                    return (ex.kind, ex.msg)
            assert run(P) is None
            assert run(P[::-1]) is None
            assert run(P[:4]) == ('InvalidSignature', '')
            assert run(P, b'wrong') == ('InvalidSignature', '')
            if mode is not None:
                other = ref.LEGACY if mode == ref.MODERN else ref.MODERN
                try:
                    S.verify_secure_with_mode(P, msg, other)
                    crossed = None
                except pkg.BlsError as ex:
                    crossed = ex.kind
                assert crossed == 'InvalidSignature'
            assert run([]) == ('InvalidSignature', '')
            ident = pkg.Signature(impl, scheme, sigraw(None))
            assert run([], sig=ident) is None
            assert run(P, sig=ident) == ('InvalidInputs', 'signature is the identity point')
        # rogue key: pk_r = x g - pk_0 and naive aggregate must fail (reference :501-540)
        x = 0x5151
        rogue = C.pk_curve.add(C.pk_curve.mul(C.pk_gen, x), C.pk_curve.neg(pks[0]))
        forged = C.sig_curve.mul(C.hash_to_point(msg, C.DST[scheme]), x)
        S = pkg.Signature(impl, scheme, sigraw(forged, rng))
        with pytest.raises(pkg.BlsError):
            S.verify_secure([pkg.PublicKey(impl, pkraw(pks[0], rng)), pkg.PublicKey(impl, pkraw(rogue, rng))], msg)


def test_verify_secure_large(api):
    """verify_secure at n = 3,000 keys made on the device; the aggregate is built from the library's own coefficient
    step, so this is a round trip (sort -> H -> t_i -> MSM -> verify) at a size the oracle would need minutes for."""
    n = 3000
    sks = [0x777 + 13 * i for i in range(n)]
    msg = b'large secure aggregate'
    pks, sigs = api.sign_batch(2, api.BASIC, sks, [msg] * n)        # G2Impl: 48-byte keys, both modes exist
    for mode in (api.MODERN, api.LEGACY):
        kb = api.serialize(1, pks, legacy=(mode == api.LEGACY))
        st, perm, ts = api.secure_coefficients(kb)
        assert st == 0 and sorted(perm) == list(range(n))
        agg = api.point_sum(2, [sigs[i] for i in perm], ts)
        assert api.verify_secure(2, api.BASIC, pks, agg, msg, mode) == api.OK
        assert api.verify_secure(2, api.BASIC, pks[:-1], agg, msg, mode) == api.INVALID_SIGNATURE
        assert api.verify_secure(2, api.BASIC, pks, agg, msg, 1 - mode) == api.INVALID_SIGNATURE


def test_verify_batch_large_property(api):
    """BASELINE-size style property at n = 8,192: device-signed batch, every 97th message tampered, every 101st
    signature replaced by its neighbour's: the verdict vector must be exactly the tamper pattern."""
    n = 8192
    sks = [0xabcdef + i for i in range(n)]
    msgs = [hashlib.sha256(i.to_bytes(8, 'little')).digest() for i in range(n)]
    pks, sigs = api.sign_batch(1, api.POP, sks, msgs)
    expect = [0] * n
    msgs2, sigs2 = list(msgs), list(sigs)
    for i in range(0, n, 97):
        msgs2[i] = msgs2[i][:-1] + bytes([msgs2[i][-1] ^ 1])
        expect[i] = 1
    for i in range(5, n, 101):
        sigs2[i] = sigs[(i + 1) % n]
        expect[i] = 1
    assert api.verify_batch(1, api.POP, pks, sigs2, msgs2) == expect


@pytest.mark.parametrize('path', ['cooperative', 'lane_split'])
@pytest.mark.parametrize('sg', [1, 2])
def test_verify_batch_vs_c_oracle(api, sg, path):
    """Device-signed items with tampered messages, swapped signatures and swapped keys, both orientations, all three schemes'
    DSTs: the status vector must equal the C oracle's, item by item.  704 items take the wave-cooperative pairing and the
    two-lane prepare, 6,656 (beyond the threshold, BLSGPU_COOP_MAX = 4,096 since round 3) the lane-split Miller / final-exponentiation kernels."""
    import os
    n = 704 if path == 'cooperative' else int(os.environ.get('BLS_DIFF_N', '6656'))     # soak runs: BLS_DIFF_N=60000
    bo = util.load_c_oracle()
    rng = random.Random(100 + sg)
    for scheme in (api.BASIC, api.AUG, api.POP):
        sks = [rng.randrange(1, 2**200) for _ in range(n)]
        msgs = [hashlib.sha256(b'co%d-%d' % (scheme, i)).digest()[:rng.randrange(1, 33)] for i in range(n)]
        pks, sigs = api.sign_batch(sg, scheme, sks, msgs)
        pks, sigs, msgs = list(pks), list(sigs), list(msgs)
        for i in range(0, n, 7):
            k = i % 21
            if k == 0:
                msgs[i] = msgs[i] + b'?'
            elif k == 7:
                sigs[i] = sigs[(i + 1) % n]
            else:
                pks[i] = pks[(i + 2) % n]
        got = api.verify_batch(sg, scheme, pks, sigs, msgs)
        blob = b''.join(msgs)
        offs = (ctypes.c_uint64 * (n + 1))()
        o = 0
        for i, m in enumerate(msgs):
            offs[i] = o
            o += len(m)
        offs[n] = o
        st = (ctypes.c_int32 * n)()
        V = lambda x: ctypes.cast(x, ctypes.c_void_p)  # noqa: E731
        bo.bo_verify_batch(sg, scheme, V(ctypes.c_char_p(b''.join(pks))), V(ctypes.c_char_p(b''.join(sigs))), V(ctypes.c_char_p(blob)), V(offs), n,
                           V(st), min(16, os.cpu_count() or 1))
        assert got == list(st), scheme
        assert sum(got) > n // 8          # the tampering took


@pytest.mark.parametrize('sg', [1, 2])
def test_verify_batch_ragged_sizes(api, sg):
    """Batch sizes around the wave (32 items) and workgroup boundaries and on both sides of the cooperative / lane-split
    threshold (4,096 items since round 3; 6,145 stays in the list as a plain lane-split size), empty batch included: one tampered item per batch, exact verdict vectors, both orientations."""
    sizes = (0, 1, 2, 31, 32, 33, 63, 65, 255, 257, 511, 512, 513, 1023, 1024, 1025, 1057, 4096, 4097, 4129, 6145) if sg == 1 else (0, 1, 33, 511, 513, 1025, 4097)
    nmax = max(sizes)
    sks = [0x5151 + 3 * i for i in range(nmax)]
    msgs = [hashlib.sha256(b'ragged%d' % i).digest() for i in range(nmax)]
    pks, sigs = api.sign_batch(sg, api.POP, sks, msgs)
    for n in sizes:
        m = list(msgs[:n])
        expect = [0] * n
        if n:
            j = (7 * n) // 11
            m[j] = m[j] + b'!'
            expect[j] = 1
        assert api.verify_batch(sg, api.POP, pks[:n], sigs[:n], m) == expect, n


@pytest.mark.parametrize('sg', [1, 2])
def test_aggregate_verify_pair_grouping(api, sg):
    """The pairing-product kernels run one Miller loop per two items: odd and even counts, including the lone tail item."""
    nmax = 35
    sks = [0x7000 + 5 * i for i in range(nmax)]
    msgs = [hashlib.sha256(b'pairs%d' % i).digest() for i in range(nmax)]
    pks, sigs = api.sign_batch(sg, api.POP, sks, msgs)
    for n in (1, 2, 3, 4, 5, 32, 33, 34, 35):
        agg = api.point_sum(sg, sigs[:n])          # signatures live in G_sg
        assert api.aggregate_verify(sg, api.POP, pks[:n], msgs[:n], agg)[0] == api.OK, n
        bad = list(msgs[:n])
        bad[n - 1] = b'tail item tampered'
        assert api.aggregate_verify(sg, api.POP, pks[:n], bad, agg)[0] == api.INVALID_SIGNATURE, n
        if n > 2:
            bad = list(msgs[:n])
            bad[(n - 1) // 2] = b'middle item tampered'
            assert api.aggregate_verify(sg, api.POP, pks[:n], bad, agg)[0] == api.INVALID_SIGNATURE, n


@pytest.mark.parametrize('group', [1, 2])
def test_msm_pippenger_closed_form(api, group):
    """Bucket-method MSM at n = 1,500 (8-bit windows) and 20,000 (11-bit windows): points sk_i * g made on the device,
    random 255-bit scalars incl. 0, 1, r-1; expected (sum t_i sk_i) * g from one oracle scalar multiplication."""
    rng = random.Random(50 + group)
    sg = 2 if group == 1 else 1                 # public keys of Bls12381G2Impl live in G1 and vice versa
    E, gen, comp = (c.E1, c.G1_GEN, c.g1_compress) if group == 1 else (c.E2, c.G2_GEN, c.g2_compress)
    for n in (1500, 20000):
        sks = [rng.randrange(1, c.R) for _ in range(n)]
        pks, _ = api.sign_batch(sg, api.BASIC, sks, [b''] * n)
        ts = [rng.randrange(c.R) for _ in range(n)]
        ts[0], ts[1], ts[2] = 0, 1, c.R - 1
        pks[7] = pks[3]                          # duplicate point
        sks[7] = sks[3]
        want = E.mul(gen, sum(t * s for t, s in zip(ts, sks)) % c.R)
        got = api.serialize(group, [api.point_sum(group, pks, ts)])[0]
        assert got == comp(want), n


@pytest.mark.parametrize('group', [1, 2])
def test_deserialize_wire_formats(api, group):
    """Checked decompression on the GPU vs the oracle: valid points, both sign bits, infinity, legacy headers, malformed
    headers, x >= p, off-curve x, on-curve points outside the subgroup (reference src/impls/legacy.rs:172-253,
    tests/legacy_comprehensive_test.rs:212-240)."""
    rng = random.Random(70 + group)
    E, gen, comp, dec = (c.E1, c.G1_GEN, c.g1_compress, c.g1_decompress) if group == 1 else (c.E2, c.G2_GEN, c.g2_compress, c.g2_decompress)
    w = 48 * group
    good = [comp(E.mul(gen, rng.randrange(1, c.R))) for _ in range(8)] + [comp(None)]
    flipped = [bytes([b[0] ^ 0x20]) + b[1:] for b in good[:8]]
    bad = [bytes(w), bytes([0x40]) + bytes(w - 1), bytes([0xc0]) + bytes(w - 2) + b'\x01', bytes([0xe0]) + bytes(w - 1),
           bytes([0x9f]) + b'\xff' * (w - 1)]
    scan = []
    x0 = rng.randrange(c.P)
    for k in range(24):
        x = (x0 + k) % c.P
        enc = x.to_bytes(48, 'big') if group == 1 else (5).to_bytes(48, 'big') + x.to_bytes(48, 'big')
        scan.append(bytes([enc[0] | 0x80]) + enc[1:])
    blobs = good + flipped + bad + scan

    def want(b):
        try:
            return comp(dec(b)), 0
        except c.DecodeError:
            return None, 7
    pts, st = api.deserialize(group, blobs)
    expect = [want(b) for b in blobs]
    assert st == [e[1] for e in expect]
    back = api.serialize(group, [p for p, s in zip(pts, st) if s == 0])
    assert back == [e[0] for e in expect if e[1] == 0]
    assert sum(1 for e in expect[-24:] if e[1] == 7) >= 5 and any(e[1] == 0 for e in expect[:9])
    # legacy headers
    leg = [ref.modern_to_legacy(b) for b in good]
    pts, st = api.deserialize(group, leg + [bytes([0xff]) * w, bytes([0x20]) + bytes(w - 1)], legacy=True)
    assert st == [0] * 9 + [8, 8]
    assert api.serialize(group, pts[:9]) == good


def test_verify_batch_wire_formats(api):
    """verify_batch straight from wire bytes (keys 48 B / signatures 96 B of Bls12381G2Impl, modern and legacy), incl. the
    reference's C++ vectors as raw bytes (tests/cpp_integration_test.rs:35-82) and undecodable items."""
    k = KATS['cpp']
    msg = bytes.fromhex(k['message'])
    pks = [bytes.fromhex(h) for h in k['pk']]
    sigs = [bytes.fromhex(h) for h in k['sig']]
    assert api.verify_batch(2, api.BASIC, pks, sigs, [msg] * 3, fmt=api.FMT_COMPRESSED) == [0, 0, 0]
    leg_p, leg_s = [ref.modern_to_legacy(b) for b in pks], [ref.modern_to_legacy(b) for b in sigs]
    assert api.verify_batch(2, api.BASIC, leg_p, leg_s, [msg] * 3, fmt=api.FMT_LEGACY) == [0, 0, 0]
    bad_pk = [pks[0], bytes(48), pks[2]]
    bad_sig = [sigs[1], sigs[1], bytes([0x40]) + bytes(95)]
    assert api.verify_batch(2, api.BASIC, bad_pk, bad_sig, [msg] * 3, fmt=api.FMT_COMPRESSED) == [1, 7, 7]
    assert api.verify_batch(2, api.BASIC, [bytes([0xff]) * 48], [leg_s[0]], [msg], fmt=api.FMT_LEGACY) == [8]
    # G1Impl orientation from wire bytes
    C = ref.G1Impl
    sk = ref.keygen_from_hash(b'\x21' * 32)
    pk, sig = C.pk_to_bytes(ref.public_key(C, sk)), C.sig_to_bytes(ref.sign(C, ref.AUG, sk, b'wire'))
    assert api.verify_batch(1, api.AUG, [pk], [sig], [b'wire'], fmt=api.FMT_COMPRESSED) == [0]
    assert api.verify_batch(1, api.AUG, [pk], [sig], [b'wirf'], fmt=api.FMT_COMPRESSED) == [1]


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_pop_verify_batch(api, C, sg):
    """ProofOfPossession::verify (reference src/proof_of_possession.rs:79-81, src/traits/sig_pop.rs:61-70)."""
    rng = random.Random(80 + sg)
    pkraw, sigraw = raw_fns(sg)
    sks, pks = keys(C, 6, 21)
    proofs = [ref.pop_prove(C, s) for s in sks]
    expect = [0] * 6
    proofs[2] = proofs[3]                      # someone else's proof
    expect[2] = 1
    pk_in = list(pks)
    pk_in[4] = None
    expect[4] = 3
    pr_in = list(proofs)
    pr_in[5] = None
    expect[5] = 2
    for i in range(6):
        try:
            ref.pop_verify(C, pk_in[i], pr_in[i])
            assert expect[i] == 0
        except ref.BlsError:
            assert expect[i] != 0
    assert api.pop_verify_batch(sg, [pkraw(p, rng) for p in pk_in], [sigraw(s, rng) for s in pr_in]) == expect
    # a signature under the signing DST is not a proof of possession
    sig = ref.sign(C, ref.POP, sks[0], C.pk_to_bytes(pks[0]))
    assert api.pop_verify_batch(sg, [pkraw(pks[0], rng)], [sigraw(sig, rng)]) == [1]


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_aggregate_secure(api, C, sg):
    """aggregate_secure[_with_mode] (reference src/secure_aggregation.rs:110-169,338-352): equals the oracle's aggregate,
    incl. duplicate keys (first matching signature), and round-trips through verify_secure; n = 2,000 uses the bucket MSM."""
    rng = random.Random(90 + sg)
    pkraw, sigraw = raw_fns(sg)
    sks, pks = keys(C, 7, 31)
    msg = b'aggregate me'
    sigs = [ref.sign(C, ref.BASIC, s, msg) for s in sks]
    pks[5], sigs[5] = pks[1], ref.sign(C, ref.BASIC, sks[1], b'another message')      # duplicate key, different signature
    for mode in ([0] if sg == 1 else [0, 1]):
        st, agg = api.aggregate_secure(sg, [pkraw(p, rng) for p in pks], [sigraw(s, rng) for s in sigs], mode)
        want = ref.aggregate_secure(C, pks, sigs, None if sg == 1 else mode)
        assert st == 0 and api.serialize(sg, [agg])[0] == C.sig_to_bytes(want)
    assert api.serialize(sg, [api.aggregate_secure(sg, [], [])[1]])[0] == C.sig_to_bytes(None)
    n = 2000
    dsks = [0x4242 + 3 * i for i in range(n)]
    dpks, dsigs = api.sign_batch(sg, api.BASIC, dsks, [msg] * n)
    st, agg = api.aggregate_secure(sg, dpks, dsigs)
    assert st == 0 and api.verify_secure(sg, api.BASIC, dpks, agg, msg) == 0
    assert api.verify_secure(sg, api.BASIC, dpks[1:], agg, msg) == 1


# ------------------------------------------------------------------ other two-pairing checks
@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_sig_proof_verify_batch(api, C, sg):
    """ProofOfKnowledge::verify (reference src/proof_of_knowledge.rs:132-164 -> src/traits/sig_proof.rs:102-142; round trip
    as in the reference's tests/proof_of_knowledge.rs): valid proofs under the three DSTs, a wrong challenge, a wrong
    message, a proof for another key, and the four InvalidInputs cases in the reference's order."""
    rng = random.Random(110 + sg)
    pkraw, sigraw = raw_fns(sg)
    sks, pks = keys(C, 3, 41)
    cases = []          # (scheme, u, v, pk, y, msg)
    for i, scheme in enumerate([ref.BASIC, ref.AUG, ref.POP]):
        msg = b'proof of knowledge %d' % i
        sig = C.sig_curve.mul(C.hash_to_point(msg, C.DST[scheme]), sks[i])     # the message is hashed as given (no key prefix)
        x, y = rng.randrange(1, c.R), rng.randrange(1, c.R)
        u, v = ref.sig_proof_generate(C, sig, msg, C.DST[scheme], x, y)
        cases.append((scheme, u, v, pks[i], y, msg))
    s0, u0, v0, pk0, y0, m0 = cases[0]
    cases += [(s0, u0, v0, pk0, (y0 + 1) % c.R, m0), (s0, u0, v0, pk0, y0, m0 + b'!'), (s0, u0, v0, pks[1], y0, m0),
              (ref.AUG, u0, v0, pk0, y0, m0),                     # right proof, wrong DST
              (s0, None, v0, pk0, y0, m0), (s0, u0, None, pk0, y0, m0), (s0, u0, v0, None, y0, m0), (s0, u0, v0, pk0, 0, m0),
              (s0, None, None, None, 0, m0)]
    # y chosen so that T = U + y * H(m) is the identity: y = -x mod r
    x1 = rng.randrange(1, c.R)
    u1 = C.sig_curve.mul(C.hash_to_point(m0, C.DST[s0]), x1)
    cases.append((s0, u1, v0, pk0, (-x1) % c.R, m0))
    want = []
    for scheme, u, v, pk, y, msg in cases:
        try:
            ref.sig_proof_verify(C, u, v, pk, y, msg, C.DST[scheme])
            want.append(None)
        except ref.BlsError as e:
            want.append((e.kind, e.msg))
    assert want[:3] == [None] * 3 and want[3:7] == [('InvalidProof', '')] * 4 and want[-1] == ('InvalidProof', '')
    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        idx = [k for k, cs in enumerate(cases) if cs[0] == scheme]
        st = api.sig_proof_verify_batch(sg, scheme, [sigraw(cases[k][1], rng) for k in idx], [sigraw(cases[k][2], rng) for k in idx],
                                        [pkraw(cases[k][3], rng) for k in idx], [cases[k][4] for k in idx], [cases[k][5] for k in idx])
        got = [api.proof_error_from_status(s) for s in st]
        assert [None if g is None else (g.kind, g.msg) for g in got] == [want[k] for k in idx]


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_signcrypt_valid_and_share_checks(api, C, sg):
    """SignCryptCiphertext::is_valid (reference src/sign_crypt_ciphertext.rs:86-101 -> src/traits/sign_crypt.rs:69-77) and
    BlsSignCrypt::verify_share (:192-207) through the generic two-pair check."""
    rng = random.Random(120 + sg)
    pkraw, sigraw = raw_fns(sg)
    sks, pks = keys(C, 2, 51)
    cts = []
    for i, scheme in enumerate([ref.BASIC, ref.AUG, ref.POP, ref.BASIC]):
        r = rng.randrange(1, c.R)
        u = C.pk_curve.mul(C.pk_gen, r)                                # U = P^r                 sign_crypt.rs:46
        v = bytes(rng.randrange(256) for _ in range(32 + 5 * i))       # V: opaque to the validity check
        w = C.sig_curve.mul(ref.signcrypt_compute_w(C, u, v, C.DST[scheme]), r)     # W = H(U || V)^r   :59
        cts.append((scheme, u, v, w))
    s0, u0, v0, w0 = cts[0]
    cts += [(s0, u0, v0[:-1] + bytes([v0[-1] ^ 1]), w0), (s0, u0, v0, cts[3][3]), (ref.POP, u0, v0, w0), (s0, None, v0, w0),
            (s0, u0, v0, None)]
    want = [ref.signcrypt_valid(C, u, v, w, C.DST[scheme]) for scheme, u, v, w in cts]
    assert want == [True] * 4 + [False] * 5
    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        idx = [k for k, ct in enumerate(cts) if ct[0] == scheme]
        got = api.signcrypt_valid_batch(sg, scheme, [pkraw(cts[k][1], rng) for k in idx], [sigraw(cts[k][3], rng) for k in idx],
                                        [cts[k][2] for k in idx])
        assert got == [want[k] for k in idx]
    # decryption shares: share_i = U * sk_i against pk_i = g * sk_i                     sign_crypt.rs:166-183,192-207
    g1of = (lambda s, p: s) if sg == 1 else (lambda s, p: p)            # which member of a (sig-group, pk-group) pair is in G1
    g2of = (lambda s, p: p) if sg == 1 else (lambda s, p: s)
    items = []
    for k, (share_sk, pk) in enumerate([(sks[0], pks[0]), (sks[1], pks[1]), (sks[0], pks[1])]):
        share = C.pk_curve.mul(u0, share_sk)
        items.append((share, pk))
    want = [ref.signcrypt_verify_share(C, sh, pk, u0, v0, w0, C.DST[s0]) for sh, pk in items]
    assert want == [True, True, False]
    h = C.sig_curve.neg(ref.signcrypt_compute_w(C, u0, v0, C.DST[s0]))
    r1, r2 = (util.g1_raw, util.g2_raw)
    got = api.pairing2_check_batch([r1(g1of(h, sh), rng) for sh, _ in items], [r2(g2of(h, sh), rng) for sh, _ in items],
                                   [r1(g1of(w0, pk), rng) for _, pk in items], [r2(g2of(w0, pk), rng) for _, pk in items])
    assert got == want


def test_pairing2_check_batch_identities_and_lane_split(api):
    """Pairing::pairing on two pairs per item (reference src/helpers.rs:41-63): pairs with an identity member contribute 1
    (both trivial -> one; exactly one trivial -> not one); 4,200 items take the lane-split kernels, checked by bilinearity."""
    rng = random.Random(131)
    g1, g2 = c.G1_GEN, c.G2_GEN
    a, b = rng.randrange(1, c.R), rng.randrange(1, c.R)
    pa, qb = c.E1.mul(g1, a), c.E2.mul(g2, b)
    pab, nq = c.E1.mul(g1, a * b % c.R), c.E2.neg(g2)
    items = [(pa, qb, pab, nq, True), (pa, qb, pa, nq, False), (None, qb, pab, None, True), (pa, None, None, None, True),
             (pa, qb, None, nq, False), (None, qb, pab, nq, False)]
    for p0, q0, p1, q1, w in items:
        assert c.pairing_product_is_one([(p0, q0), (p1, q1)]) == w
    got = api.pairing2_check_batch([util.g1_raw(t[0], rng) for t in items], [util.g2_raw(t[1], rng) for t in items],
                                   [util.g1_raw(t[2], rng) for t in items], [util.g2_raw(t[3], rng) for t in items])
    assert got == [t[4] for t in items]
    # e(k g1, g2) * e(-g1, k g2) == 1 for device-made points; every 7th item gets a different scalar on one side
    n = 4200
    ks = [0x1357 + 11 * i for i in range(n)]
    k2 = [k + (1 if i % 7 == 3 else 0) for i, k in enumerate(ks)]
    g2pts, _ = api.sign_batch(1, api.BASIC, ks, [b''] * n)              # pk = k * g2 (RAW_PROJ)
    g1pts, _ = api.sign_batch(2, api.BASIC, k2, [b''] * n)              # pk = k * g1
    ng1 = util.g1_raw(c.E1.neg(g1))
    got = api.pairing2_check_batch(g1pts, [util.g2_raw(g2)] * n, [ng1] * n, g2pts)
    assert got == [i % 7 != 3 for i in range(n)]


def test_concurrent_callers(api):
    """The C ABI is blocking and thread-safe (threading row): four host threads issue different calls at once
    (ctypes releases the GIL) and every result equals the one obtained sequentially."""
    import threading
    n = 300
    jobs = []
    for t in range(4):
        sg = 1 + t % 2
        sks = [0x9000 + 17 * t + i for i in range(n)]
        msgs = [b'thread %d item %d' % (t, i) for i in range(n)]
        pks, sigs = api.sign_batch(sg, api.POP, sks, msgs)
        bad = list(msgs)
        for i in range(t, n, 7):
            bad[i] = b'x' + bad[i]
        jobs.append((sg, pks, sigs, bad, [1 if i % 7 == t else 0 for i in range(n)]))
    seq = [api.verify_batch(sg, api.POP, pks, sigs, m) for sg, pks, sigs, m, _ in jobs]
    assert seq == [e for *_, e in jobs]
    out = [None] * 4

    def run(k):
        sg, pks, sigs, m, _ = jobs[k]
        res = []
        for _ in range(3):
            res.append(api.verify_batch(sg, api.POP, pks, sigs, m))
            res.append(api.multi_verify(sg, api.POP, pks[:50], sigs[0], m[0]))
        out[k] = res
    th = [threading.Thread(target=run, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(4):
        sg, pks, sigs, m, e = jobs[k]
        single = api.multi_verify(sg, api.POP, pks[:50], sigs[0], m[0])
        assert out[k] == [e, single] * 3


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_empty_message_through_the_overlapped_tails(api, C, sg):
    """The zero-length message through MultiSignature::verify (message hashed on the side stream), verify_secure (message
    hashed while the host derives the coefficients) and a single verify: the oracle's verdicts."""
    rng = random.Random(140 + sg)
    pkraw, sigraw = raw_fns(sg)
    sks, pks = keys(C, 3, 61)
    for msg in (b'', b'x'):
        msig = ref.aggregate_signatures(C, [ref.sign(C, ref.POP, s, msg) for s in sks])
        ref.multi_sig_verify(C, ref.POP, pks, msig, msg)
        assert api.multi_verify(sg, api.POP, [pkraw(p, rng) for p in pks], sigraw(msig, rng), msg) == 0
        assert api.multi_verify(sg, api.POP, [pkraw(p, rng) for p in pks[:2]], sigraw(msig, rng), msg) == 1
        agg = ref.aggregate_secure(C, pks, [ref.sign(C, ref.BASIC, s, msg) for s in sks])
        ref.verify_secure(C, ref.BASIC, pks, agg, msg)
        assert api.verify_secure(sg, api.BASIC, [pkraw(p, rng) for p in pks], sigraw(agg, rng), msg) == 0
        assert api.verify_secure(sg, api.BASIC, [pkraw(p, rng) for p in pks[:2]], sigraw(agg, rng), msg) == 1
        assert api.verify_batch(sg, api.POP, [pkraw(pks[0], rng)], [sigraw(ref.sign(C, ref.POP, sks[0], msg), rng)], [msg]) == [0]
