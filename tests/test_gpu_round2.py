"""GPU parity for what round 2 moved onto the device: the key sort / coefficient step of hash_public_keys_with_sorted,
Basic's duplicate-message rule, the first-identity reduction and the neutral records of the sharded aggregate verify,
scalars >= r in the MSM, the context pool.  Everything goes through the C ABI and is compared with the oracle or with a
plain Python restatement of the reference's loop."""
import ctypes
import os
import hashlib
import random

import pytest

import util
from util import c, ref

pytestmark = pytest.mark.gpu


def _ops(pkg):
    import torch
    return pkg.api.TensorOps(torch.device('cuda', 0))


def _t(ops, b):
    import torch
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(ops.device) if b else torch.zeros(0, dtype=torch.uint8, device=ops.device)


# ------------------------------------------------------------------ hash_public_keys_with_sorted on the device
@pytest.mark.parametrize('width', [48, 96])
def test_device_key_sort_is_stable_and_lexicographic(api, pkg, width):
    """The sort of reference src/secure_aggregation.rs:41-44 / :273-276 (sort_by on the serialised bytes, stable): random keys,
    keys that share their first 8 bytes (the prefix pass ties and the full-width pass must decide), exact duplicates (input
    order must be kept), sizes around the 64-lane and 1,024-key tile edges."""
    ops = _ops(pkg)
    rng = random.Random(width)
    for n in (1, 2, 63, 64, 65, 1023, 1024, 1025, 5000):
        for flavour in ('random', 'shared_prefix', 'duplicates'):
            keys = [rng.randbytes(width) for _ in range(n)]
            if flavour == 'shared_prefix' and n > 2:
                pre = keys[0][:8]
                for i in range(0, n, 3):
                    keys[i] = pre + keys[i][8:]
                keys[n // 2] = pre + keys[1][8:40] + keys[n // 2][40:]
            if flavour == 'duplicates' and n > 2:
                for i in range(2, n, 5):
                    keys[i] = keys[i % 2]
            kb = _t(ops, b''.join(keys))
            perm = [int(x) for x in ops.sort_keys(kb, n, width).cpu().tolist()]
            assert perm == sorted(range(n), key=lambda i: keys[i]), (n, flavour)
            first = [int(x) for x in ops.first_occurrence(kb, ops.sort_keys(kb, n, width), n, width).cpu().tolist()]
            seen = {}
            for i, k in enumerate(keys):
                seen.setdefault(k, i)
            assert first == [seen[keys[g]] for g in perm], (n, flavour)
            # H and t_i of the shard [base, base + count) in input order
            dig = ops.keys_digest(kb, ops.sort_keys(kb, n, width), n, width)
            H = hashlib.sha256(b''.join(keys[g] for g in perm)).digest()
            assert bytes(dig.cpu().numpy().tobytes()) == H
            if n <= 1025:
                base, count = n // 3, n - n // 3 - n // 5
                scal, st = ops.coefficients_for_range(dig, ops.sort_keys(kb, n, width), n, base, count)
                got = bytes(scal.cpu().numpy().tobytes())
                pos = {g: p for p, g in enumerate(perm)}
                for g in range(base, base + count):
                    t = int.from_bytes(hashlib.sha256(pos[g].to_bytes(4, 'big') + H).digest(), 'big') % c.R
                    assert got[32 * (g - base):32 * (g - base + 1)] == t.to_bytes(32, 'little')
                assert st == 0


def test_secure_coefficients_large_matches_oracle(api):
    """blsgpu_secure_coefficients at a size that spans many sort tiles, against the oracle's restatement of
    reference src/secure_aggregation.rs:41-103 (33 % of the coefficient hashes are >= r and must be reduced)."""
    rng = random.Random(5)
    n = 20000
    keys = [rng.randbytes(48) for _ in range(n)]
    keys[777] = keys[12]
    st, perm, ts = api.secure_coefficients(keys)
    wperm, _, wts = ref.secure_coefficients(keys)
    assert st == 0 and perm == wperm and ts == wts


# ------------------------------------------------------------------ Basic's duplicate-message rule on the device
def _first_dup(msgs):
    seen = {}
    for i, m in enumerate(msgs):       # reference src/traits/sig_basic.rs:46-58
        if m in seen:
            return seen[m], i
        seen[m] = i
    return None


def test_first_duplicate_message(api):
    rng = random.Random(3)
    cases = [[], [b''], [b'', b''], [b'a', b'b', b'a', b'b'], [b'x' * 100, b'x' * 99, b'x' * 100]]
    for n in (64, 65, 1000, 30000):
        msgs = [hashlib.sha256(i.to_bytes(4, 'big')).digest()[:rng.randrange(0, 33)] + bytes([i & 255, (i >> 8) & 255, i >> 16]) for i in range(n)]
        cases.append(list(msgs))                       # all distinct
        d = list(msgs)
        d[n - 1] = d[3]
        d[n // 2] = d[n // 2 - 1]                       # two duplicate pairs: the one with the smaller SECOND index wins
        cases.append(d)
        cases.append([b'same'] * n)                    # every message equal: (0, 1)
    for msgs in cases:
        assert api.first_duplicate_message(msgs) == _first_dup(msgs), len(msgs)


@pytest.mark.parametrize('sg', [1, 2])
def test_aggregate_verify_device_pointers(api, pkg, sg):
    """blsgpu_aggregate_verify with every buffer resident on the device: duplicate rule, identity reduction and verdict all
    stay on the device; the results equal the host-pointer path's (and the oracle's error precedence, reference
    src/traits/sig_basic.rs:46-58 before src/traits/sig_core.rs:155-167)."""
    import torch
    ops = _ops(pkg)
    lib = ops.lib
    n = 70
    sks = [1000 + 7 * i for i in range(n)]
    msgs = [b'msg-%d' % i for i in range(n)]
    pks, sigs = api.sign_batch(sg, api.BASIC, sks, msgs)
    agg = api.point_sum(sg, sigs)
    ident_pk = (util.g2_raw if sg == 1 else util.g1_raw)(None)
    ident_sig = (util.g1_raw if sg == 1 else util.g2_raw)(None)

    def run_dev(pk_rows, msg_rows, sig):
        offs, t = [0], 0
        for m in msg_rows:
            t += len(m)
            offs.append(t)
        d_pks, d_msgs, d_sig = _t(ops, b''.join(pk_rows)), _t(ops, b''.join(msg_rows)), _t(ops, sig)
        d_offs = torch.tensor(offs, dtype=torch.int64, device=ops.device)
        d_st = torch.full((1,), -9, dtype=torch.int32, device=ops.device)
        d_aux = torch.zeros(2, dtype=torch.int64, device=ops.device)
        api._check(lib.blsgpu_aggregate_verify(sg, api.BASIC, ops._p(d_pks), ops._p(d_msgs), ops._p(d_offs), len(pk_rows), ops._p(d_sig), 0,
                                               ops._p(d_st), ops._p(d_aux)))
        return int(d_st.item()), tuple(int(x) for x in d_aux.tolist())

    variants = {'ok': (pks, msgs, agg)}
    bad = list(msgs)
    bad[5] = b'tampered'
    variants['bad'] = (pks, bad, agg)
    dup = list(msgs)
    dup[66], dup[40] = dup[2], dup[39]
    variants['dup'] = (pks, dup, agg)
    pid = list(pks)
    pid[65], pid[9] = ident_pk, ident_pk
    variants['pkid'] = (pid, msgs, agg)
    variants['sigid'] = (pid, msgs, ident_sig)
    variants['dup_and_ids'] = (pid, dup, ident_sig)
    want = {'ok': (0, (0, 0)), 'bad': (1, (0, 0)), 'dup': (4, (39, 40)), 'pkid': (3, (10, 0)), 'sigid': (2, (0, 0)), 'dup_and_ids': (4, (39, 40))}
    for name, (p_, m_, s_) in variants.items():
        assert run_dev(p_, m_, s_) == want[name], name
        assert api.aggregate_verify(sg, api.BASIC, p_, m_, s_) == want[name], name


# ------------------------------------------------------------------ sharded aggregate verify: neutral records, no-signature shards
@pytest.mark.parametrize('sg', [1, 2])
def test_aggregate_partial_edge_shards(api, sg):
    """blsgpu_aggregate_partial for the shards the world-2 tests cannot produce: an EMPTY shard with and without the
    signature, a shard without the signature whose size is not a multiple of the wave (lane n must stay idle), and folding the
    neutral record; the product over the shards is the verdict of the unsharded call."""
    C = ref.G1Impl if sg == 1 else ref.G2Impl
    pkraw, sigraw = (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)
    rng = random.Random(sg)
    n = 7
    sks = [ref.keygen_from_hash(bytes([i, 9]) * 16) for i in range(n)]
    pks = [pkraw(ref.public_key(C, s), rng) for s in sks]
    msgs = [b'shard %d' % i for i in range(n)]
    asig = sigraw(ref.aggregate_signatures(C, [ref.sign(C, ref.BASIC, s, m) for s, m in zip(sks, msgs)]), rng)
    one = util.f12_record(c.F12_ONE)
    rec_e, fb = api.aggregate_partial(sg, api.BASIC, [], [], None)
    assert (rec_e, fb) == (one, -1)
    rec_s, fb = api.aggregate_partial(sg, api.BASIC, [], [], asig)          # the signature pair alone
    assert fb == -1 and rec_s != one
    rec_i, fb = api.aggregate_partial(sg, api.BASIC, [], [], sigraw(None))  # identity signature on an empty shard: index n = 0
    assert fb == 0
    recs = [rec_s, rec_e]
    for lo, hi in ((0, 3), (3, 3), (3, 7)):                                 # shards WITHOUT the signature, one of them empty
        r, fb = api.aggregate_partial(sg, api.BASIC, pks[lo:hi], msgs[lo:hi], None)
        assert fb == -1
        recs.append(r)
    assert api.fp12_product_is_one(recs)
    assert not api.fp12_product_is_one(recs[1:])
    # an identity key inside a shard: its index comes back, the record still folds (the pair contributes 1)
    pid = list(pks)
    pid[5] = pkraw(None)
    r, fb = api.aggregate_partial(sg, api.BASIC, pid[3:7], msgs[3:7], None)
    assert fb == 2


def test_init_rejects_a_different_device(api):
    lib = api.load_library()
    assert lib.blsgpu_init(-1) == 0 and lib.blsgpu_init(0) == 0
    assert lib.blsgpu_init(5) == -3                    # BLSGPU_E_ARG: already bound to device 0
    buf = ctypes.create_string_buffer(256)
    lib.blsgpu_last_error(buf, 256)
    assert b'already bound' in buf.value


@pytest.mark.parametrize('group', [1, 2])
def test_msm_scalars_beyond_the_group_order(api, group):
    """blsgpu_msm takes scalars modulo r on BOTH of its paths (double-and-add below 1,024 points, windows above): scalars
    with bit 255 set and scalars in [r, 2^256) give the same group element as their residues."""
    rng = random.Random(group + 40)
    E, gen, comp = (c.E1, c.G1_GEN, c.g1_compress) if group == 1 else (c.E2, c.G2_GEN, c.g2_compress)
    for n in (1023, 1024):
        ks = [3 + i for i in range(n)]
        pts, _ = api.sign_batch(3 - group, api.POP, ks, [b'x'] * n)          # pk_i = (3 + i) g of the wanted group
        scal = [rng.randrange(c.R) for _ in range(n)]
        scal[0] = 2 ** 256 - 1
        scal[1] = c.R
        scal[2] = c.R + 5
        scal[3] = 2 ** 255
        scal[n - 1] = 2 ** 255 + 12345
        total = sum(k * (s % c.R) for k, s in zip(ks, scal)) % c.R
        got = api.serialize(group, [api.point_sum(group, pts, scal)])[0]
        assert got == comp(E.mul(gen, total)), n


def test_context_pool_overlaps_callers(api):
    """Two host threads inside the library at the same time lease different contexts: both calls return the sequential
    results (tests/test_gpu_api.py::test_concurrent_callers checks four threads; here the point is that a long call does
    not block a short one behind one mutex -- the short call finishes while the long one is still running)."""
    import threading
    import time
    n = 20000
    pks, sigs = api.sign_batch(1, api.POP, [5 + i for i in range(n)], [b'm'] * n)
    one_pk, one_sig = pks[:1], sigs[:1]
    api.verify_batch(1, api.POP, one_pk, one_sig, [b'm'])
    t_long = {}

    def long_call():
        t0 = time.perf_counter()
        st = api.verify_batch(1, api.POP, pks, sigs, [b'm'] * n)
        t_long['dt'] = time.perf_counter() - t0
        t_long['ok'] = all(s == 0 for s in st)

    th = threading.Thread(target=long_call)
    th.start()
    time.sleep(0.002)
    t0 = time.perf_counter()
    st = api.verify_batch(1, api.POP, one_pk, one_sig, [b'm'])
    dt_short = time.perf_counter() - t0
    th.join()
    assert st == [0] and t_long['ok']
    assert dt_short < t_long['dt']


def test_hand_over_kernels_under_a_full_device(api):
    """The single-verdict checks whose workgroups wait for each other inside one launch (k_pairing_post2, k_pairing_stream) while
    another caller keeps every CU busy: a 65,536-item batch holds all wave slots and nearly all LDS, so the engine workgroups of the
    short calls are placed one by one as room appears -- a consumer must never be resident without its producer having been
    placed (they are the lower block indices), and no bounded wait may run out (status -2).  Three threads of short calls -- one
    Bls12381G1Impl verification, one Bls12381G2Impl verification, a MultiSignature::verify tail -- alternate valid and tampered inputs
    for a few seconds beside the batches; every verdict is the sequential one."""
    import threading
    import time
    n = 65536
    pks, sigs = api.sign_batch(1, api.POP, [77 + i for i in range(n)], [b'full device'] * n)
    small = {}
    for sg in (1, 2):
        small[sg] = api.sign_batch(sg, api.POP, [991, 992], [b'short call', b'short call'])
    mk, ms = api.sign_batch(1, api.POP, [3000 + i for i in range(200)], [b'tail'] * 200)
    magg = api.point_sum(1, ms)
    stop = time.perf_counter() + 6.0
    bad, counts = [], {'batch': 0, 1: 0, 2: 0, 'multi': 0}

    def batches():
        while time.perf_counter() < stop:
            st = api.verify_batch(1, api.POP, pks, sigs, [b'full device'] * n)
            if any(st):
                bad.append(('batch', [x for x in st if x][:3]))
            counts['batch'] += 1

    def singles(sg):
        pk, sig = small[sg]
        k = 0
        while time.perf_counter() < stop:
            want = k & 1
            got = api.verify_batch(sg, api.POP, [pk[want]], [sig[0]], [b'short call'])[0]     # key 1 with signature 0: invalid
            if got != want:
                bad.append((sg, k, got))
            k += 1
        counts[sg] = k

    def multi():
        k = 0
        while time.perf_counter() < stop:
            want = k & 1
            got = api.multi_verify(1, api.POP, mk[:200 - want], magg, b'tail')
            if got != want:
                bad.append(('multi', k, got))
            k += 1
        counts['multi'] = k

    th = [threading.Thread(target=batches), threading.Thread(target=singles, args=(1,)), threading.Thread(target=singles, args=(2,)),
          threading.Thread(target=multi)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]
    assert counts['batch'] >= 2 and counts[1] >= 20 and counts[2] >= 20 and counts['multi'] >= 20, counts


# ------------------------------------------------------------------ N2: the serde_bare form of Signature<C>
@pytest.mark.parametrize('C,sg', [(ref.G1Impl, 1), (ref.G2Impl, 2)], ids=['g1', 'g2'])
def test_signature_serde_bare_tagged_bytes(api, C, sg):
    """Vec<u8>::from(&Signature) / Signature::try_from(&[u8]) (reference src/signature.rs:112-126): scheme tag byte + compressed
    point; the reference's own test asserts the lengths 49 / 97 and the round trip for the three schemes (:279-318)."""
    rng = random.Random(sg)
    sk = ref.keygen_from_hash(b'test_try_from')
    msg = b'test_try_from'
    sigraw = util.g1_raw if sg == 1 else util.g2_raw
    sig_pts = [ref.sign(C, scheme, sk, msg) for scheme in (ref.BASIC, ref.AUG, ref.POP)]
    raws = [sigraw(p, rng) for p in sig_pts]
    blobs = api.signatures_to_tagged(sg, [0, 1, 2], raws)
    assert [len(b) for b in blobs] == [49 if sg == 1 else 97] * 3
    assert blobs == [bytes([scheme]) + C.sig_to_bytes(p) for scheme, p in zip((0, 1, 2), sig_pts)]
    schemes, pts, sts = api.signatures_from_tagged(sg, blobs)
    assert schemes == [0, 1, 2] and sts == [0, 0, 0]
    assert api.serialize(sg, pts) == [C.sig_to_bytes(p) for p in sig_pts]
    # the decoded signatures verify under their own scheme
    pk = (util.g2_raw if sg == 1 else util.g1_raw)(ref.public_key(C, sk), rng)
    for scheme, p in zip(schemes, pts):
        assert api.verify_batch(sg, scheme, [pk], [p], [msg]) == [0]
    # malformed records: unknown tag, corrupted point, wrong length
    bad_tag = bytes([3]) + blobs[0][1:]
    bad_point = blobs[1][:1] + bytes([blobs[1][1] ^ 0x80]) + blobs[1][2:]
    _, _, sts = api.signatures_from_tagged(sg, [bad_tag, bad_point, blobs[2][:-1], blobs[2]])
    assert sts == [api.BAD_ENCODING, api.BAD_ENCODING, api.BAD_LENGTH, 0]


# ------------------------------------------------------------------ N3: PublicKeyShare::verify
@pytest.mark.parametrize('C,sg', [(ref.G1Impl, 1), (ref.G2Impl, 2)], ids=['g1', 'g2'])
def test_public_key_share_verify(api, C, sg):
    """PublicKeyShare::verify (reference src/public_key_share.rs:53-72) is the scheme's verify on the share VALUES: Shamir shares
    f(1..5) of a secret (threshold 3), each partial signature checked against its public-key share in one batch; a share
    paired with another participant's signature fails; the Lagrange combination of three partial signatures verifies under
    the group key (what the partial checks protect)."""
    rng = random.Random(31 + sg)
    pkraw, sigraw = (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)
    coef = [rng.randrange(1, c.R) for _ in range(3)]
    share = [sum(a * pow(x, k, c.R) for k, a in enumerate(coef)) % c.R for x in range(1, 6)]
    msg = b'threshold message'
    for scheme in (ref.BASIC, ref.AUG, ref.POP):
        pk_shares = [ref.public_key(C, s) for s in share]
        sig_shares = [ref.sign(C, scheme, s, msg) for s in share]
        praw, sraw = [pkraw(p, rng) for p in pk_shares], [sigraw(s, rng) for s in sig_shares]
        assert api.verify_batch(sg, scheme, praw, sraw, [msg] * 5) == [0] * 5
        swapped = [sraw[1], sraw[0]] + sraw[2:]
        assert api.verify_batch(sg, scheme, praw, swapped, [msg] * 5) == [1, 1, 0, 0, 0]
        if scheme != ref.AUG:     # Aug signs pk_share || msg: partial signatures do not combine across different prefixes
            xs = [1, 3, 4]
            lam = []
            for i in xs:
                num = den = 1
                for j in xs:
                    if j != i:
                        num = num * j % c.R
                        den = den * (j - i) % c.R
                lam.append(num * pow(den, -1, c.R) % c.R)
            comb = api.point_sum(sg, [sraw[x - 1] for x in xs], lam)
            group_pk = pkraw(ref.public_key(C, coef[0]), rng)
            assert api.verify_batch(sg, scheme, [group_pk], [comb], [msg]) == [0]


def test_in_library_multi_device(tmp_path):
    """blsgpu_init_devices: one process driving several devices (here three logical devices on the one GPU of the box,
    BLSGPU_FAKE_DEVICES) shards the four verify entry points inside the library -- contiguous ranges, one host thread per
    device, partial key sums / Fp12 Miller products / MSM partials folded on device 0 -- and must return exactly what the
    single-device library returns: verdict vectors, error precedence and indices, for host and for device-resident inputs."""
    import json
    import os
    import subprocess
    import sys
    out = str(tmp_path / 'multidev.json')
    env = dict(os.environ)
    env.pop('BLSGPU_FAKE_DEVICES', None)
    subprocess.check_call([sys.executable, os.path.join(util.ROOT, 'tests', 'multidev_worker.py'), out], env=env, timeout=600)
    res = json.load(open(out))
    assert res['equal'], {k: (res['single'][k], res['multi'][k]) for k in res['single'] if res['single'][k] != res['multi'][k]}
    s = res['single']
    for sg in (1, 2):
        assert [i for i, v in enumerate(s['vb_%d' % sg]) if v] == [5, 60, 150] and s['vb_dev_%d' % sg] == s['vb_%d' % sg]
        assert s['mv_%d' % sg] == [0, 1, 1]
        assert s['av_%d' % sg] == [[0, [0, 0]], [1, [0, 0]], [4, [7, 180]], [1, [0, 0]], [3, [34, 0]], [2, [0, 0]]]
        assert s['av_dev_%d' % sg] == [1, [0, 0]]
        for mode in ([0] if sg == 1 else [0, 1]):
            assert s['vs_%d_%d' % (sg, mode)] == [0, 0, 1, 1]


@pytest.mark.parametrize('group', [1, 2])
def test_point_sum_tail_on_the_engine(api, group):
    """the last levels of every point sum run on the row-wide engine (k_point_tree_wide: sixteen points per workgroup, complete
    projective additions, csrc/wide_tables.cuh set PT): sums that cancel to the identity, identities among and instead of the
    inputs, inputs that repeat (the doubling case of the addition law), sizes around the sixteen-point groups"""
    from oracle.py import bls381 as c
    rng = random.Random(40 + group)
    E, gen, raw, comp = (c.E1, c.G1_GEN, util.g1_raw, c.g1_compress) if group == 1 else (c.E2, c.G2_GEN, util.g2_raw, c.g2_compress)
    base = [E.mul(gen, rng.randrange(1, c.R)) for _ in range(6)]

    def check(pts):
        want = None
        for q in pts:
            want = E.add(want, q)
        got = api.serialize(group, [api.point_sum(group, [raw(q, rng) for q in pts])])[0]
        assert got == comp(want), len(pts)

    check([base[0], E.neg(base[0])])                                   # cancels in the first addition
    check([None] * 40)                                                  # nothing but identities
    check([base[k % 3] for k in range(48)] + [E.neg(base[k % 3]) for k in range(48)])   # cancels only at the top of the tree
    check([base[0]] * 256)                                              # every addition is a doubling
    for n in (15, 16, 17, 31, 33, 255, 257, 1023, 1025):
        pts = [base[rng.randrange(6)] if rng.random() < 0.8 else None for _ in range(n)]
        check(pts)


@pytest.mark.parametrize('group', [1, 2])
def test_point_sum_of_deserialised_keys(api, group):
    """keys with Z = 1 (what decoding the wire bytes yields) and RAW_AFF inputs take the mixed addition in the first stage of a
    point sum: same group element as the oracle's serial sum, incl. a repeated key (the doubling branch of the mixed adder), a
    key and its negative, identities, and Z = 1 mixed with arbitrary Z in one call"""
    from oracle.py import bls381 as c
    rng = random.Random(60 + group)
    E, gen, raw, comp = (c.E1, c.G1_GEN, util.g1_raw, c.g1_compress) if group == 1 else (c.E2, c.G2_GEN, util.g2_raw, c.g2_compress)
    aff_raw, aff_sz = (util.g1_aff_raw, 96) if group == 1 else (util.g2_aff_raw, 192)
    base = [E.mul(gen, rng.randrange(1, c.R)) for _ in range(5)]
    for n in (1, 2, 7, 64, 65, 700):
        pts = [base[rng.randrange(5)] for _ in range(n)]
        if n >= 7:
            pts[3] = E.neg(pts[2])
            pts[5] = None
        want = None
        for q in pts:
            want = E.add(want, q)
        got = api.serialize(group, [api.point_sum(group, [raw(q) for q in pts])])[0]                       # RAW_PROJ, Z = 1
        assert got == comp(want), n
        mixed = [raw(q) if k % 2 else raw(q, rng) for k, q in enumerate(pts)]                               # Z = 1 next to random Z
        assert api.serialize(group, [api.point_sum(group, mixed)])[0] == comp(want), n
        affs = [aff_raw(q) if q is not None else bytes(aff_sz) for q in pts]                                # RAW_AFF (identity = zeros)
        assert api.serialize(group, [api.point_sum(group, affs, fmt=api.FMT_RAW_AFFINE)])[0] == comp(want), n


def test_grouped_verification_matches_per_item(api):
    """blsgpu_verify_batch_grouped (opt-in: groups of eight items share one final exponentiation through a seeded random linear
    combination, failing groups are re-verified item by item) returns the status vector of blsgpu_verify_batch: tampered
    messages, swapped signatures, identity signatures and keys, sizes around the group boundaries, all three schemes, several
    seeds, and a batch with no and with only invalid items."""
    rng = random.Random(808)
    for scheme in (api.BASIC, api.AUG, api.POP):
        for n in (1, 7, 8, 9, 16, 41, 200):
            sks = [rng.randrange(1, 2 ** 250) for _ in range(n)]
            msgs = [bytes([rng.randrange(256) for _ in range(rng.randrange(0, 70))]) for _ in range(n)]
            pks, sigs = api.sign_batch(1, scheme, sks, msgs)
            pks, sigs, msgs = list(pks), list(sigs), list(msgs)
            for i in range(n):
                roll = rng.random()
                if roll < 0.10:
                    msgs[i] = msgs[i] + b'!'                                  # tampered message
                elif roll < 0.15 and n > 1:
                    sigs[i] = sigs[(i + 1) % n]                               # someone else's signature
                elif roll < 0.18:
                    sigs[i] = util.g1_raw(None)                               # identity signature
                elif roll < 0.21:
                    pks[i] = util.g2_raw(None)                                # identity key
            want = api.verify_batch(1, scheme, pks, sigs, msgs)
            for seed in (1, 0xdeadbeef):
                assert api.verify_batch_grouped(1, scheme, pks, sigs, msgs, seed) == want, (scheme, n, seed)
    # all valid (no fallback at all) and all invalid (every group falls back)
    n = 64
    sks = [1000 + i for i in range(n)]
    msgs = [b'm%d' % i for i in range(n)]
    pks, sigs = api.sign_batch(1, api.POP, sks, msgs)
    assert api.verify_batch_grouped(1, api.POP, pks, sigs, msgs) == [0] * n
    assert api.verify_batch_grouped(1, api.POP, pks, sigs, [m + b'x' for m in msgs]) == [1] * n
    assert api.verify_batch_grouped(1, api.POP, [], [], []) == []
    with pytest.raises(api.BlsGpuRuntimeError):
        api.verify_batch_grouped(2, api.POP, pks, sigs, msgs)                 # built for Bls12381G1Impl only


def test_grouped_verification_rejects_the_cancelling_forgery(api):
    """The advisor's round-2 finding: when the scalars of the random linear combination depend only on a public seed and the item
    index, sig_a + [r_b] D and sig_b - [r_a] D cancel in a group's combined check and two forged signatures pass.  The scalars are
    now derived from the group's own inputs (csrc/kernels.cuh grouped_scalar), so the forgery built for the old seed-only
    scalars -- and the same forgery with arbitrary scalars -- is reported invalid, item by item, exactly as blsgpu_verify_batch
    reports it."""
    rng = random.Random(4242)
    M64 = (1 << 64) - 1

    def old_scalar(seed, i):                       # round 2's splitmix64(seed, i) | 1
        z = (seed + 0x9e3779b97f4a7c15 * (i + 1)) & M64
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
        return (z ^ (z >> 31)) | 1

    n = 8
    sks = [rng.randrange(1, c.R) for _ in range(n)]
    msgs = [b'grouped-%d' % i for i in range(n)]
    C = ref.G1Impl
    pts = [ref.sign(C, ref.POP, sk, m) for sk, m in zip(sks, msgs)]
    pks = [util.g2_raw(ref.public_key(C, sk), rng) for sk in sks]
    D = c.E1.mul(c.G1_GEN, rng.randrange(1, c.R))
    for seed in (1, 0x626c73677075):               # bench.py's first seed and api.py's former default
        for ra, rb in ((old_scalar(seed, 2), old_scalar(seed, 5)), (rng.randrange(1, 1 << 64), rng.randrange(1, 1 << 64))):
            forged = list(pts)
            forged[2] = c.E1.add(pts[2], c.E1.mul(D, rb))
            forged[5] = c.E1.add(pts[5], c.E1.neg(c.E1.mul(D, ra)))
            sigs = [util.g1_raw(p, rng) for p in forged]
            want = [0] * n
            want[2] = want[5] = api.INVALID_SIGNATURE
            assert api.verify_batch(1, api.POP, pks, sigs, msgs) == want
            assert api.verify_batch_grouped(1, api.POP, pks, sigs, msgs, seed) == want, (seed, ra, rb)
    # the scalars depend on the inputs: the same seed, two different batches -> both verdict vectors still exact
    assert api.verify_batch_grouped(1, api.POP, pks, [util.g1_raw(p, rng) for p in pts], msgs, 1) == [0] * n


def test_raw_points_outside_the_prime_order_subgroups(api):
    """The contract edge of the RAW formats (include/blsgpu.h; VERDICT r2 weak #1, advisor's low finding): the reference can only hold
    subgroup members (from_bytes checks, src/public_key.rs:58-74), a RAW caller could hand over anything on the curve.  Pinned:
      * a RAW signature with a cofactor-torsion component, sig + T1 (T1 = [r] R1 != 0): the verdict is that of sig -- T1 lies in
        r E1(Fp), where the reduced pairing is trivial -- in the cleared form (core_verify on hashed points, pair (sig, -g2)) and in
        the uncleared form the batch paths run (pair (sig, -[c] g2)), on every size class of verify_batch;
      * a RAW key outside G2, pk + T2: INVALID_SIGNATURE (the Miller function of a point outside the eigenspace; the oracle's
        pairing of the same raw points agrees), never accepted;
      * the wire formats reject both with BAD_ENCODING, as the reference's from_bytes does."""
    rng = random.Random(2024)
    C = ref.G1Impl

    def torsion(curve, sqrt, k):                                       # a non-trivial point of the cofactor torsion: [r] (random curve point)
        while True:
            x = rng.randrange(util.P) if k == 1 else (rng.randrange(util.P), rng.randrange(util.P))
            y = sqrt(curve.rhs(x))
            if y is None:
                continue
            t = curve.mul((x, y), c.R)
            if t is not None:
                return t

    T1 = torsion(c.E1, c.fp_sqrt, 1)
    T2 = torsion(c.E2, c.f2_sqrt, 2)
    assert c.E1.on_curve(T1) and c.E2.on_curve(T2) and not c.g1_in_subgroup(T1) and not c.g2_in_subgroup(T2)
    for n in (1, 40, 600, 7000):                                        # the cut check, the engine, the wave-cooperative and the lane-split kernels
        sks = [rng.randrange(1, c.R) for _ in range(4)]
        msgs = [b'edge-%d' % i for i in range(n)]
        pk_pts = [ref.public_key(C, sks[i % 4]) for i in range(n)]
        sig_pts = [ref.sign(C, ref.POP, sks[i % 4], msgs[i]) if i < 8 or i % 97 == 0 else None for i in range(n)]
        pks, sigs, want = [], [], []
        for i in range(n):
            if sig_pts[i] is None:                                      # filler: identity signature (status 2), cheap to build
                pks.append(util.g2_raw(pk_pts[i], rng)); sigs.append(util.g1_raw(None)); want.append(api.SIG_IDENTITY)
            elif i % 3 == 0:                                            # signature with a torsion component: still valid
                pks.append(util.g2_raw(pk_pts[i], rng)); sigs.append(util.g1_raw(c.E1.add(sig_pts[i], T1), rng)); want.append(api.OK)
            elif i % 3 == 1:                                            # key outside G2: invalid
                pks.append(util.g2_raw(c.E2.add(pk_pts[i], T2), rng)); sigs.append(util.g1_raw(sig_pts[i], rng)); want.append(api.INVALID_SIGNATURE)
            else:
                pks.append(util.g2_raw(pk_pts[i], rng)); sigs.append(util.g1_raw(sig_pts[i], rng)); want.append(api.OK)
        assert api.verify_batch(1, api.POP, pks, sigs, msgs) == want, n
    # the oracle's pairing on the same raw points gives the same verdicts
    m = b'edge-oracle'
    pk, sig = ref.public_key(C, 77), ref.sign(C, ref.POP, 77, m)
    H = c.hash_to_g1(m, C.DST[ref.POP])
    neg_g2 = c.E2.neg(c.G2_GEN)
    assert c.pairing_product_is_one([(H, pk), (c.E1.add(sig, T1), neg_g2)])
    assert not c.pairing_product_is_one([(H, c.E2.add(pk, T2)), (sig, neg_g2)])
    # (the cleared form, pair (sig, -g2), on the same signature: tests/test_hostsim.py::test_verify_items, CPU)
    # wire formats: both points are refused at the door
    bad_sig, bad_pk = c.g1_compress(c.E1.add(sig, T1)), c.g2_compress(c.E2.add(pk, T2))
    assert api.verify_batch(1, api.POP, [c.g2_compress(pk), bad_pk], [bad_sig, c.g1_compress(sig)], [m, m], fmt=api.FMT_COMPRESSED) == [api.BAD_ENCODING] * 2


# ------------------------------------------------------------------ hash-to-curve paths of different batch sizes
@pytest.mark.parametrize('group', [1, 2])
def test_hash_to_point_paths_agree_and_match_the_c_restatement(api, group):
    """HashToPoint::hash_to_point (reference src/impls/g1.rs:17-19, g2.rs:15-17): up to 128 messages take one workgroup each
    (row-wide SSWU maps, the rest on the engine), more take the one-wave / lane-pair kernels: both against oracle/c on ragged
    message lengths (empty, one SHA block, across block edges), and against each other on the shared prefix of the batch."""
    rng = random.Random(77 + group)
    bo = util.load_c_oracle()
    lens = [0, 1, 31, 32, 54, 55, 56, 63, 64, 65, 119, 120, 127, 128, 129, 200, 257]
    msgs = [rng.randbytes(lens[i % len(lens)]) for i in range(140)]
    dst = b'BLS_SIG_BLS12381G%d_XMD:SHA-256_SSWU_RO_POP_' % group
    small = api.serialize(group, api.hash_to_point(group, msgs[:128], dst))
    large = api.serialize(group, api.hash_to_point(group, msgs, dst))
    assert large[:128] == small
    width = 48 if group == 1 else 96
    for i in list(range(0, 140, 7)) + [127, 128, 139]:
        out = ctypes.create_string_buffer(width)
        bo.bo_hash_to_point(group, msgs[i], len(msgs[i]), dst, len(dst), out)
        assert large[i] == out.raw, i


# ------------------------------------------------------------------ a short leg of the randomised campaign inside the suite
def test_randomised_campaign_short(api):
    """tests/stress_parity.py (random sizes across every path boundary, tamperings, schemes, both implementations; verify_batch,
    hash-to-curve, MultiSignature::verify, verify_secure, AggregateSignature::verify against oracle/c) for a fixed seed and a
    bounded number of rounds; the long runs are recorded in profiles/r02_stress_parity.txt."""
    import stress_parity as sp
    bo = util.load_c_oracle()
    rng = random.Random(20261004)
    kinds = (sp.round_verify_batch, sp.round_hash, sp.round_multi, sp.round_aggregate)
    for k in range(40):
        kind = kinds[k % 4] if k < 8 else rng.choices(kinds, (5, 2, 2, 3))[0]
        line = []
        assert kind(api, bo, rng, line.append), (k, line)

@pytest.mark.gpu
def test_segmented_final_exponentiation_and_one_kernel_miller_loop_agree():
    """The A/B forms kept beside the defaults -- BLSGPU_FINALEXP_SEG=1 (six segments + the four-lanes-per-item compressed squarings of
    k_cyc_run4), BLSGPU_FINALEXP_V1=1 and BLSGPU_MILLER_V1=1 (rounds 1-2's one-kernel final exponentiation / Miller loop) -- return the
    default path's verdict vector on a lane-split batch with tampered items and identities (child processes: the switches are read once)."""
    import subprocess
    import sys
    code = (
        "import sys, random; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge, util\n"
        "api = ge.import_pkg().api; api.init()\n"
        "n = 6400; rng = random.Random(5)\n"
        "sks = [1000 + i for i in range(n)]; msgs = [b'seg-%%d' %% i for i in range(n)]\n"
        "out = []\n"
        "for sg in (1, 2):\n"
        "    pks, sigs = api.sign_batch(sg, api.POP, sks, msgs)\n"
        "    pks, sigs, ms = list(pks), list(sigs), list(msgs)\n"
        "    for i in range(0, n, 37): ms[i] = ms[i] + b'!'\n"
        "    sigs[11] = util.g1_raw(None) if sg == 1 else util.g2_raw(None)\n"
        "    out.append(api.verify_batch(sg, api.POP, pks, sigs, ms))\n"
        "print(repr(out))\n") % (util.ROOT, os.path.join(util.ROOT, 'tests'))
    res = {}
    for name, env in (('default', {}), ('seg', {'BLSGPU_FINALEXP_SEG': '1'}), ('fe_v1', {'BLSGPU_FINALEXP_V1': '1'}), ('miller_v1', {'BLSGPU_MILLER_V1': '1'})):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, BLSGPU_AB_KNOBS='1', **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        res[name] = eval(r.stdout.strip().splitlines()[-1])
    want = res['default']
    for sg_idx in (0, 1):
        exp = [1 if i % 37 == 0 else 0 for i in range(6400)]
        exp[11] = 2
        assert want[sg_idx] == exp
    for name in ('seg', 'fe_v1', 'miller_v1'):
        assert res[name] == want, name


@pytest.mark.gpu
def test_verify_secure_with_and_without_weighted_tables():
    """verify_secure's key sum with every key's weighted window multiples built under the host hash (k_msm2_tables, the default) and
    without them (BLSGPU_MSM2_TABLES=0: the chunk lanes double their sums into place): the same verdicts for a valid aggregate, a key
    short and a wrong message, both orientations, at a size whose windows differ in width (child processes: the knob is read once)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "api = ge.import_pkg().api; api.init()\n"
        "out = []\n"
        "for sg, kg in ((1, 2), (2, 1)):\n"
        "    n = 1500\n"
        "    sks = [0x4242 + 7 * i for i in range(n)]; msg = b'tables or not'\n"
        "    pks, sigs = api.sign_batch(sg, api.BASIC, sks, [msg] * n)\n"
        "    st, perm, ts = api.secure_coefficients(api.serialize(kg, pks))\n"
        "    agg = api.point_sum(sg, [sigs[i] for i in perm], ts)\n"
        "    out.append([st, api.verify_secure(sg, api.BASIC, pks, agg, msg), api.verify_secure(sg, api.BASIC, pks[:-1], agg, msg),\n"
        "                api.verify_secure(sg, api.BASIC, pks, agg, msg + b'!')])\n"
        "print(repr(out))\n") % (util.ROOT, os.path.join(util.ROOT, 'tests'))
    res = {}
    for name, env in (('tables', {}), ('plain', {'BLSGPU_MSM2_TABLES': '0'})):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        res[name] = eval(r.stdout.strip().splitlines()[-1])
    assert res['tables'] == [[0, 0, 1, 1], [0, 0, 1, 1]]
    assert res['plain'] == res['tables']


@pytest.mark.gpu
def test_streamed_split_and_plain_cut_checks_agree():
    """The single-verdict checks in the forms that hand data between workgroups of one launch, against the plain ones: the checks whose
    lines cannot be had early -- the summed key of MultiSignature::verify / verify_secure (Bls12381G1Impl), H(m) of a Bls12381G2Impl
    verification -- with the lines travelling while the Miller loop runs (k_pairing_stream; BLSGPU_STREAM_LINES=0: two launches), and
    the late part of every other cut check with its Miller loop on three or two workgroups (k_pairing_post2; BLSGPU_POST_SPLIT=2: two, 0: one).  Both
    are the default up to 64 / 128 items: the same verdicts for valid and tampered inputs at one item, at the forms' upper bounds and just
    beyond them, both orientations, repeated on one context (the hand-over flags are reused with a fresh value per launch).  Child
    processes: the knobs are read once."""
    import subprocess
    import sys
    code = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "api = ge.import_pkg().api; api.init()\n"
        "out = []\n"
        "n = 300; sks = [0x5151 + 5 * i for i in range(n)]; msg = b'lines beside the loop'\n"
        "pks, sigs = api.sign_batch(1, api.POP, sks, [msg] * n)\n"
        "agg = api.point_sum(1, sigs)\n"
        "for rep in range(3):\n"
        "    out.append([api.multi_verify(1, api.POP, pks, agg, msg), api.multi_verify(1, api.POP, pks[:-1], agg, msg),\n"
        "                api.multi_verify(1, api.POP, pks, agg, msg + b'?')])\n"
        "pkb, sgb = api.sign_batch(1, api.BASIC, sks, [msg] * n)\n"
        "st, perm, ts = api.secure_coefficients(api.serialize(2, pkb))\n"
        "sagg = api.point_sum(1, [sgb[i] for i in perm], ts)\n"
        "out.append([st, api.verify_secure(1, api.BASIC, pkb, sagg, msg), api.verify_secure(1, api.BASIC, pkb[1:], sagg, msg)])\n"
        "pk2m, sg2m = api.sign_batch(2, api.POP, sks, [msg] * n)\n"
        "agg2 = api.point_sum(2, sg2m)\n"
        "out.append([api.multi_verify(2, api.POP, pk2m, agg2, msg), api.multi_verify(2, api.POP, pk2m[:-1], agg2, msg)])\n"
        "for sg in (2, 1):\n"
        "    for m in (1, 3, 64, 65, 128, 129):\n"
        "        ms = [hashlib.sha256(b'item %%d' %% i).digest() for i in range(m)]\n"
        "        pkm, sgm = api.sign_batch(sg, api.POP, sks[:m], ms)\n"
        "        ms[m // 2] = ms[m // 2] + b'!'\n"
        "        out.append(list(api.verify_batch(sg, api.POP, pkm, sgm, ms)))\n"
        "print(repr(out))\n") % (util.ROOT, os.path.join(util.ROOT, 'tests'))
    res = {}
    variants = (('default', {}), ('two_launches', {'BLSGPU_STREAM_LINES': '0'}), ('two_workgroups', {'BLSGPU_POST_SPLIT': '2'}),
                ('one_workgroup', {'BLSGPU_POST_SPLIT': '0'}), ('plain', {'BLSGPU_STREAM_LINES': '0', 'BLSGPU_POST_SPLIT': '0'}))
    for name, env in variants:
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        res[name] = eval(r.stdout.strip().splitlines()[-1])
    want = [[0, 1, 1]] * 3 + [[0, 0, 1], [0, 1]]
    for sg in (2, 1):
        for m in (1, 3, 64, 65, 128, 129):
            want.append([1 if i == m // 2 else 0 for i in range(m)])
    for name, _ in variants:
        assert res[name] == want, name


@pytest.mark.gpu
def test_pairing_product_tree_and_accumulator_forms_agree(api):
    """AggregateSignature::verify's pairing product in its two forms -- the per-entry products over the items with one Horner chain
    (k_line_quad / k_f12_fold4 / k_f12_horner_wide, the default from 64 pairs) and one accumulator per item or pair of items
    (BLSGPU_PRODUCT_TREE=0: k_millerfp3 / k_millerfp) -- give the same verdicts, the same 576-byte Miller-product records for shards
    without the signature's pair and records that fold together for shards with it (blsgpu_aggregate_partial) at sizes around the form's lower bound, the fold levels'
    fan-in (4, 16) and with flagged (identity) keys, tampered messages and a missing signature pair (child processes: the switch is read once)."""
    import subprocess
    import sys
    code = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge, util\n"
        "api = ge.import_pkg().api; api.init()\n"
        "out = []\n"
        "for sg in (1, 2):\n"
        "    nmax = 4200\n"
        "    sks = [0x9000 + 3 * i for i in range(nmax)]; msgs = [hashlib.sha256(b'tree%%d' %% i).digest() for i in range(nmax)]\n"
        "    pks, sigs = api.sign_batch(sg, api.BASIC, sks, msgs)\n"
        "    for n in (62, 63, 64, 65, 67, 255, 256, 257, 1030, 4100):\n"
        "        agg = api.point_sum(sg, sigs[:n])\n"
        "        out.append(api.aggregate_verify(sg, api.BASIC, pks[:n], msgs[:n], agg))\n"
        "        bad = list(msgs[:n]); bad[n // 3] = b'tampered'\n"
        "        out.append(api.aggregate_verify(sg, api.BASIC, pks[:n], bad, agg))\n"
        "        ident = list(pks[:n]); ident[n - 2] = util.g2_raw(None) if sg == 1 else util.g1_raw(None)\n"
        "        out.append(api.aggregate_verify(sg, api.BASIC, ident, msgs[:n], agg))\n"
        "        out.append(api.aggregate_partial(sg, api.BASIC, pks[:n], msgs[:n], agg))\n"
        "        out.append(api.aggregate_partial(sg, api.BASIC, ident, msgs[:n], None))\n"
        "print(repr(out))\n") % (util.ROOT, os.path.join(util.ROOT, 'tests'))
    res = {}
    variants = (('tree', {}), ('acc', {'BLSGPU_PRODUCT_TREE': '0'}),
                # the per-entry form's own switches: the two-lane line kernel everywhere; chunk-local fold levels from 64 values on and
                # the engine only for the last sixteen; k_prepare_agg on one lane per item
                ('lines2', {'BLSGPU_LINES4_MAX': '0'}), ('levels', {'BLSGPU_TREE_LOCAL': '64', 'BLSGPU_TREE_ENGINE_FROM': '16'}),
                ('agg1', {'BLSGPU_AGG_LANES': '1'}))
    for name, env in variants:
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, BLSGPU_AB_KNOBS='1', **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        res[name] = eval(r.stdout.strip().splitlines()[-1])
    assert len(res['tree']) == 2 * 10 * 5
    for k in range(0, len(res['tree']), 5):
        for name in ('tree', 'acc'):
            ok, tampered, ident, rec, rec_ident = res[name][k:k + 5]
            assert ok[0] == 0 and tampered[0] == 1, (name, k)
            assert ident[0] != 0, (name, k)          # an identity key is an error of its own (reference src/traits/sig_core.rs:149-178)
            assert len(rec[0]) == 576 and rec[1] == -1 and rec_ident[1] >= 0, (name, k)
            # the record of the whole valid aggregate (signature's pair included) is a Miller product whose final exponentiation is 1:
            # checked HERE, by this process's library, for the records of both children
            assert api.fp12_product_is_one([rec[0]]), (name, k)
            assert not api.fp12_product_is_one([rec_ident[0]]), (name, k)
        # without the signature's pair the two forms multiply the same line values: byte-identical records.  With it the per-entry
        # form takes that pair's lines from the NORMALISED table of the fixed argument (every row divided by a factor in Fp2, which
        # the final exponentiation removes): the records differ by such a factor and fold together all the same (above)
        assert res['acc'][k + 4] == res['tree'][k + 4], k
        assert [r[0] for r in res['acc'][k:k + 3]] == [r[0] for r in res['tree'][k:k + 3]], k
    # the switches of the per-entry form change its plan, never a result
    for name in ('lines2', 'levels', 'agg1'):
        assert [r[0] for r in res[name]] == [r[0] for r in res['tree']], name
        for k in range(3, len(res['tree']), 5):
            assert api.fp12_product_is_one([res[name][k][0]]) and res[name][k + 1] == res['tree'][k + 1], (name, k)
