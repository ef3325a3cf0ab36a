"""The plain-C oracle (oracle/c): pinned by the reference's known-answer vectors and cross-checked against oracle/py."""
import ctypes
import json
import os
import random
import subprocess

import pytest

import util
from util import c, ref


@pytest.fixture(scope='module')
def bo():
    return util.load_c_oracle()


def test_c_oracle_reference_kats(bo):
    """K2 (C++ signatures verify, tests/cpp_integration_test.rs:103-104), K3 (naive aggregate fails verify_secure, :171-191)
    and K4 (57-signer production vector, tests/secure_aggregation_test.rs:143-235)."""
    k = json.load(open(os.path.join(util.ROOT, 'tests', 'golden', 'ref_kats.json')))
    C = ref.G2Impl
    cpp = k['cpp']
    msg = bytes.fromhex(cpp['message'])
    pks = [C.pk_from_bytes(bytes.fromhex(h)) for h in cpp['pk']]
    sigs = [C.sig_from_bytes(bytes.fromhex(h)) for h in cpp['sig']]
    for pk, sig in zip(pks, sigs):
        assert bo.bo_verify(2, 0, util.g1_raw(pk), util.g2_raw(sig), msg, len(msg)) == 0
        assert bo.bo_verify(2, 0, util.g1_raw(pk), util.g2_raw(sig), b'hellp', 5) == 1
    naive = C.sig_from_bytes(bytes.fromhex(cpp['naive_agg_sig_pk12']))
    assert bo.bo_verify_secure(2, 0, b''.join(util.g1_raw(p) for p in pks[:2]), 2, util.g2_raw(naive), msg, len(msg), 0) == 1
    for n in (2, 3):
        agg = ref.aggregate_secure(C, pks[:n], sigs[:n])
        assert bo.bo_verify_secure(2, 0, b''.join(util.g1_raw(p) for p in pks[:n]), n, util.g2_raw(agg), msg, len(msg), 0) == 0
    p57 = k['prod57']
    pk57 = b''.join(util.g1_raw(C.pk_from_bytes(bytes.fromhex(h))) for h in p57['pks'])
    sig57 = util.g2_raw(C.sig_from_bytes(bytes.fromhex(p57['sig'])))
    m57 = bytes.fromhex(p57['message'])
    assert bo.bo_verify_secure(2, 0, pk57, 57, sig57, m57, len(m57), 0) == 0
    assert bo.bo_verify_secure(2, 0, pk57, 57, sig57, m57, len(m57), 1) == 1          # cross-mode fails
    assert bo.bo_verify_secure(2, 0, pk57, 56, sig57, m57, len(m57), 0) == 1


def test_c_oracle_vs_python_oracle(bo):
    rng = random.Random(3)
    for group, comp, h in ((1, c.g1_compress, c.hash_to_g1), (2, c.g2_compress, c.hash_to_g2)):
        for m in (b'', b'abc', bytes(range(130))):
            dst = b'QUUX-V01-CS02-with-BLS12381G%d_XMD:SHA-256_SSWU_RO_' % group
            out = ctypes.create_string_buffer(48 * group)
            bo.bo_hash_to_point(group, m, len(m), dst, len(dst), out)
            assert out.raw == comp(h(m, dst))
        for dl in (255, 256, 300):          # RFC 9380 5.3.3: from 256 bytes on the DST is hashed first (pinned by K.2 in test_oracle.py)
            dst = (b'oversize-dst-%d-' % group) * 30
            bo.bo_hash_to_point(group, b'abc', 3, dst[:dl], dl, out)
            assert out.raw == comp(h(b'abc', dst[:dl])), dl
    for C, sg in ((ref.G1Impl, 1), (ref.G2Impl, 2)):
        pkraw, sigraw = (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)
        for scheme in (ref.BASIC, ref.AUG, ref.POP):
            sk = ref.keygen_from_hash(bytes([scheme + 40 * sg]) * 32)
            pk = ref.public_key(C, sk)
            m = b'signatures_work'
            sig = ref.sign(C, scheme, sk, m)
            assert bo.bo_verify(sg, scheme, pkraw(pk, rng), sigraw(sig, rng), m, len(m)) == 0
            assert bo.bo_verify(sg, scheme, pkraw(pk, rng), sigraw(sig, rng), b'bad', 3) == 1
            assert bo.bo_verify(sg, scheme, pkraw(pk, rng), sigraw(None), m, len(m)) == 2
            assert bo.bo_verify(sg, scheme, pkraw(None), sigraw(sig, rng), m, len(m)) == 3
            assert bo.bo_verify(sg, scheme, pkraw(None), sigraw(None), m, len(m)) == 2
        # threaded batch entry
        n = 12
        items = []
        for i in range(n):
            sk = ref.keygen_from_hash(bytes([i, sg]) * 16)
            m = b'm%d' % i
            items.append((ref.public_key(C, sk), ref.sign(C, ref.POP, sk, m), m if i % 3 else m + b'!'))
        offs = (ctypes.c_uint64 * (n + 1))()
        blob = b''
        for i, it in enumerate(items):
            offs[i] = len(blob)
            blob += it[2]
        offs[n] = len(blob)
        st = (ctypes.c_int32 * n)()
        pkb, sgb = b''.join(pkraw(it[0], rng) for it in items), b''.join(sigraw(it[1], rng) for it in items)
        P = lambda x: ctypes.cast(x, ctypes.c_void_p)  # noqa: E731
        bo.bo_verify_batch(sg, 2, P(ctypes.c_char_p(pkb)), P(ctypes.c_char_p(sgb)), P(ctypes.c_char_p(blob)), P(offs), n, P(st), 4)
        assert list(st) == [0 if i % 3 else 1 for i in range(n)]
    # legacy-mode verify_secure vs the Python oracle (G2Impl, 5 signers)
    C = ref.G2Impl
    sks = [ref.keygen_from_hash(bytes([i + 90]) * 32) for i in range(5)]
    pks = [ref.public_key(C, s) for s in sks]
    msg = b'legacy'
    sigs = [ref.sign(C, ref.BASIC, s, msg) for s in sks]
    for mode in (ref.MODERN, ref.LEGACY):
        agg = ref.aggregate_secure(C, pks, sigs, mode)
        raw = b''.join(util.g1_raw(p, rng) for p in pks)
        assert bo.bo_verify_secure(2, 0, raw, 5, util.g2_raw(agg, rng), msg, len(msg), mode) == 0
        assert bo.bo_verify_secure(2, 0, raw, 5, util.g2_raw(agg, rng), msg, len(msg), 1 - mode) == 1
    assert bo.bo_verify_secure(2, 0, b'', 0, util.g2_raw(None), msg, len(msg), 0) == 0
    assert bo.bo_verify_secure(2, 0, b'', 0, util.g2_raw(sigs[0]), msg, len(msg), 0) == 1


def test_c_oracle_config_legs_vs_python_oracle(bo):
    """The CPU legs of BASELINE configs 3-5 (bo_multi_verify, bo_aggregate_verify, bo_verify_secure_mt), single-threaded =
    the reference's loops and multi-threaded, against oracle/py: verdicts, error precedence and indices."""
    rng = random.Random(8)
    V = lambda b: ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p)  # noqa: E731
    for C, sg in ((ref.G1Impl, 1), (ref.G2Impl, 2)):
        pkraw, sigraw = (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)
        n = 11                      # > 8 pairs per thread: the multi-pair Miller loop allocates its point table
        sks = [ref.keygen_from_hash(bytes([i, sg, 3]) * 10 + b'xx') for i in range(n)]
        pks = [ref.public_key(C, s) for s in sks]
        praw = b''.join(pkraw(p, rng) for p in pks)
        m1 = b'one message'
        msig = sigraw(ref.aggregate_signatures(C, [ref.sign(C, ref.POP, s, m1) for s in sks]), rng)
        for th in (1, 3):
            assert bo.bo_multi_verify(sg, 2, V(praw), n, V(msig), m1, len(m1), th) == 0
            assert bo.bo_multi_verify(sg, 2, V(praw), n - 1, V(msig), m1, len(m1), th) == 1
        msgs = [b'item %d' % i for i in range(n)]

        def agg(scheme, pk_rows, msg_rows, sig, th):
            offs = (ctypes.c_uint64 * (len(msg_rows) + 1))()
            t = 0
            for i, m in enumerate(msg_rows):
                offs[i] = t
                t += len(m)
            offs[len(msg_rows)] = t
            aux = (ctypes.c_uint64 * 2)()
            st = bo.bo_aggregate_verify(sg, scheme, V(b''.join(pk_rows)), V(b''.join(msg_rows)), ctypes.cast(offs, ctypes.c_void_p), len(msg_rows), V(sig), th,
                                        ctypes.cast(aux, ctypes.c_void_p))
            return st, (aux[0], aux[1])
        rows = [pkraw(p, rng) for p in pks]
        for scheme in (ref.BASIC, ref.AUG):
            asig = sigraw(ref.aggregate_signatures(C, [ref.sign(C, scheme, s, m) for s, m in zip(sks, msgs)]), rng)
            for th in (1, 2, 4):
                assert agg(scheme, rows, msgs, asig, th) == (0, (0, 0))
                bad = list(msgs)
                bad[4] = b'tampered'
                assert agg(scheme, rows, bad, asig, th) == (1, (0, 0))
                dup = list(msgs)
                dup[6], dup[3] = dup[1], dup[2]
                assert agg(scheme, rows, dup, asig, th) == ((4, (2, 3)) if scheme == ref.BASIC else (1, (0, 0)))
                pid = list(rows)
                pid[5] = pid[2] = pkraw(None)
                assert agg(scheme, pid, msgs, asig, th) == (3, (3, 0))
                assert agg(scheme, pid, msgs, sigraw(None), th) == (2, (0, 0))
        ssigs = [C.sig_curve.mul(C.hash_to_point(m1, C.DST[ref.BASIC]), s) for s in sks]
        for mode in ([0] if sg == 1 else [0, 1]):
            sagg = sigraw(ref.aggregate_secure(C, pks, ssigs, None if sg == 1 else mode), rng)
            for th in (1, 3):
                assert bo.bo_verify_secure_mt(sg, 0, V(praw), n, V(sagg), m1, len(m1), mode, th) == 0
                assert bo.bo_verify_secure_mt(sg, 0, V(praw), n - 1, V(sagg), m1, len(m1), mode, th) == 1


def test_c_oracle_threaded_legs_under_sanitizers(tmp_path):
    """oracle/c/san_driver.c: the threaded CPU legs (config 4's 16,384-pair bo_aggregate_verify on 16 threads and on one, its error
    precedence, bo_multi_verify, bo_verify_secure[_mt] incl. the legacy transcode, a threaded bo_verify_batch) built with
    -fsanitize=address,undefined -fno-sanitize-recover: no finding, and the expected status codes.  Added after a round-2 run of
    tools/bench_configs.py dumped core in its config-4 section without an explanation: not reproducible --
    the committed code is clean at the sizes of that run; the 1-thread leg at the full 16,384 pairs (2 minutes under ASan) was
    run by hand with the same result."""
    exe = str(tmp_path / 'san_driver')
    src = os.path.join(util.ROOT, 'oracle', 'c', 'san_driver.c')
    subprocess.check_call(['gcc', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-Wall', '-Wno-unused-function',
                           '-o', exe, src, '-lpthread'])
    r = subprocess.run([exe, '16384', '16'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'san_driver ok' in r.stdout
    r = subprocess.run([exe, '512', '1'], capture_output=True, text=True, timeout=600)       # the same legs inline on the calling thread
    assert r.returncode == 0, r.stdout + r.stderr
