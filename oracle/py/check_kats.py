"""Pin the oracle against every known-answer vector the reference's own tests hold.

ORACLE self-check (test infrastructure).  Data: tests/golden/ref_kats.json (extracted by
tests/golden/make_ref_kats.py from tests/cpp_integration_test.rs and tests/secure_aggregation_test.rs).
Run:  python -m oracle.py.check_kats
"""
import json
import os
import time

from . import bls381 as c
from . import blsful_ref as ref

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = os.path.join(HERE, '..', '..', 'tests', 'golden', 'ref_kats.json')


def run(verbose=True):
    k = json.load(open(KATS))
    C = ref.G2Impl
    cpp = k['cpp']
    msg = bytes.fromhex(cpp['message'])
    pks, sigs = [], []
    for sk_h, pk_h, sig_h in zip(cpp['sk'], cpp['pk'], cpp['sig']):
        sk = int.from_bytes(bytes.fromhex(sk_h), 'big')          # SecretKey::try_from = from_be_bytes
        pk = C.pk_from_bytes(bytes.fromhex(pk_h))
        # K1: sk -> pk  (tests/cpp_integration_test.rs:99-100,141-143)
        assert C.pk_to_bytes(ref.public_key(C, sk)).hex() == pk_h, 'K1'
        sig = C.sig_from_bytes(bytes.fromhex(sig_h))
        # K2: C++ signatures verify under Basic (:103-104,146-148)
        ref.verify(C, ref.BASIC, pk, sig, msg)
        # K2': the signature pins hash-to-G2 itself: sig == sk * H(msg)
        assert C.sig_to_bytes(ref.sign(C, ref.BASIC, sk, msg)).hex() == sig_h, 'K2 hash-to-G2'
        pks.append(pk)
        sigs.append(sig)
    # secure aggregation of the C++ keys round-trips for 2 and 3 signers (:106-121,150-165)
    for n in (2, 3):
        agg = ref.aggregate_secure(C, pks[:n], sigs[:n])
        ref.verify_secure(C, ref.BASIC, pks[:n], agg, msg)
    # K3: naive aggregate must fail verify_secure (:171-191)
    naive = C.sig_from_bytes(bytes.fromhex(cpp['naive_agg_sig_pk12']))
    assert C.sig_to_bytes(ref.aggregate_signatures(C, sigs[:2])) == bytes.fromhex(cpp['naive_agg_sig_pk12'])
    try:
        ref.verify_secure(C, ref.BASIC, pks[:2], naive, msg)
        raise AssertionError('K3')
    except ref.BlsError as e:
        assert e == ref.InvalidSignature
    # K4: 57-signer production vector (tests/secure_aggregation_test.rs:146-234)
    p57 = k['prod57']
    t0 = time.time()
    pks57 = [ref.pk_from_bytes_with_mode(C, bytes.fromhex(h), ref.MODERN) for h in p57['pks']]
    sig57 = ref.sig_from_bytes_with_mode(C, bytes.fromhex(p57['sig']), ref.MODERN)
    assert C.sig_to_bytes(sig57).hex() == p57['sig']
    ref.verify_secure(C, ref.BASIC, pks57, sig57, bytes.fromhex(p57['message']))
    _, _, ts = ref.secure_coefficients([bytes.fromhex(h) for h in p57['pks']])
    hashes_ge_r = 0
    import hashlib
    perm, H, _ = ref.secure_coefficients([bytes.fromhex(h) for h in p57['pks']])
    for i in range(57):
        if int.from_bytes(hashlib.sha256(i.to_bytes(4, 'big') + H).digest(), 'big') >= c.R:
            hashes_ge_r += 1
    if verbose:
        print('K1 K2 K3 K4 pass; 57-key vector: %d/57 coefficient hashes >= r (reduce semantics required); %.1fs'
              % (hashes_ge_r, time.time() - t0))
    return True


if __name__ == '__main__':
    run()
