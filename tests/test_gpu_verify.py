"""GPU parity: Signature::verify through the C ABI vs the oracle (reads like reference tests/signatures.rs:10-36)."""
import json
import os
import random

import pytest

import util
from util import ref

pytestmark = pytest.mark.gpu

IMPLS = [(ref.G1Impl, 1), (ref.G2Impl, 2)]


def raw_fns(sg):
    return (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_signatures_work(api, C, sg):
    """sign -> verify ok / bad message err, all three schemes (reference tests/signatures.rs:10-36)."""
    rng = random.Random(100 + sg)
    pkraw, sigraw = raw_fns(sg)
    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        sk = ref.keygen_from_hash(bytes([scheme + 3 * sg]) * 32)
        pk = ref.public_key(C, sk)
        sig = ref.sign(C, scheme, sk, b'signatures_work')
        for msg, want in ((b'signatures_work', api.OK), (b'bad message', api.INVALID_SIGNATURE)):
            st = api.verify_batch(sg, scheme, [pkraw(pk, rng)], [sigraw(sig, rng)], [msg])
            assert st == [want]
        assert api.verify_batch(sg, scheme, [pkraw(pk, rng)], [sigraw(None)], [b'x']) == [api.SIG_IDENTITY]
        assert api.verify_batch(sg, scheme, [pkraw(None)], [sigraw(sig, rng)], [b'x']) == [api.PK_IDENTITY]
        assert api.verify_batch(sg, scheme, [pkraw(None)], [sigraw(None)], [b'x']) == [api.SIG_IDENTITY]


def test_cpp_vectors(api):
    """The C++ (relic) signatures of reference tests/cpp_integration_test.rs:54-82 verify (asserted there :103-104)."""
    k = json.load(open(os.path.join(util.ROOT, 'tests', 'golden', 'ref_kats.json')))['cpp']
    C = ref.G2Impl
    msg = bytes.fromhex(k['message'])
    pks = [util.g1_raw(C.pk_from_bytes(bytes.fromhex(h))) for h in k['pk']]
    sigs = [util.g2_raw(C.sig_from_bytes(bytes.fromhex(h))) for h in k['sig']]
    assert api.verify_batch(2, api.BASIC, pks, sigs, [msg] * 3) == [0, 0, 0]
    assert api.verify_batch(2, api.BASIC, pks, sigs[1:] + sigs[:1], [msg] * 3) == [1, 1, 1]


@pytest.mark.parametrize('C,sg', IMPLS, ids=['g1', 'g2'])
def test_batch_mixed_against_oracle(api, C, sg):
    """A ragged batch (different lengths, empty message, tampered items, identities) item by item vs the oracle."""
    rng = random.Random(7 + sg)
    pkraw, sigraw = raw_fns(sg)
    n = 96 if sg == 1 else 40
    scheme = ref.POP
    pks, sigs, msgs, expect = [], [], [], []
    for i in range(n):
        sk = ref.keygen_from_hash(i.to_bytes(4, 'big') * 8)
        pk = ref.public_key(C, sk)
        m = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 31, 32, 33, 55, 56, 64, 100, 200])))
        sig = ref.sign(C, scheme, sk, m)
        kind = i % 8
        if kind == 3:
            m = m + b'\x00'
        elif kind == 5:
            sig = C.sig_curve.add(sig, C.sig_curve.mul(sig, 2))
        elif kind == 6 and i % 16 == 6:
            sig = None
        elif kind == 7 and i % 16 == 7:
            pk = None
        try:
            ref.verify(C, scheme, pk, sig, m)
            want = api.OK
        except ref.BlsError as e:
            want = api.INVALID_SIGNATURE if e.kind == 'InvalidSignature' else (
                api.SIG_IDENTITY if 'signature' in e.msg else api.PK_IDENTITY)
        pks.append(pkraw(pk, rng))
        sigs.append(sigraw(sig, rng))
        msgs.append(m)
        expect.append(want)
    assert api.verify_batch(sg, scheme, pks, sigs, msgs) == expect
    assert api.verify_batch(sg, scheme, [], [], []) == []
