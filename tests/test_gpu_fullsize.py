"""BASELINE.json's full sizes for configs 2, 3, 4 and 5, checked
through size-independent properties: inputs are signed on the device from consecutive secret keys, so the expected
aggregates are closed forms computed here with Python integers only -- the oracle could not finish these sizes.
One valid case and one minimally tampered case per config; exact status codes."""
import ctypes
import hashlib

import pytest

import util

pytestmark = pytest.mark.gpu

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
SEED = hashlib.sha256(b'blsgpu-bench-v1').digest()
S0 = int.from_bytes(SEED, 'big') % R
M_FIXED = hashlib.sha256(SEED + b'fixed').digest()


@pytest.fixture(scope='module')
def gpu(api):
    import torch
    lib = api.init()
    dev = torch.device('cuda', 0)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731

    def sign(sg, scheme, sks, msgs):
        n = len(sks)
        skb = b''.join(s.to_bytes(32, 'little') for s in sks)
        d_msgs = torch.frombuffer(bytearray(b''.join(msgs) or b'\x00'), dtype=torch.uint8).to(dev)
        offs, o = [0], 0
        for m in msgs:
            o += len(m)
            offs.append(o)
        d_offs = torch.tensor(offs, dtype=torch.int64, device=dev)
        pksz, sgsz = (288, 144) if sg == 1 else (144, 288)
        d_pks = torch.empty(n * pksz, dtype=torch.uint8, device=dev)
        d_sigs = torch.empty(n * sgsz, dtype=torch.uint8, device=dev)
        api._check(lib.blsgpu_sign_batch(sg, scheme, api._ptr(skb), P(d_msgs), P(d_offs), n, P(d_pks), P(d_sigs)))
        torch.cuda.synchronize()
        return d_pks, d_sigs, d_msgs, d_offs

    return dict(torch=torch, lib=lib, dev=dev, P=P, sign=sign)


def _i32(g):
    return g['torch'].full((1,), -9, dtype=g['torch'].int32, device=g['dev'])


def test_config3_multi_verify_1m_keys(api, gpu):
    """MultiSignature::verify over 1,048,576 G2 keys pk_i = (s0 + i) g2: sum_i sk_i = n s0 + n(n-1)/2, so the valid
    multi-signature is that scalar times H(m) -- one device signature with the summed key."""
    g, n = gpu, 1 << 20
    lib, P = g['lib'], g['P']
    d_pks, _, _, _ = g['sign'](1, api.POP, [(S0 + i) % R or 1 for i in range(n)], [b''] * n)
    _, d_sig, _, _ = g['sign'](1, api.POP, [(n * S0 + n * (n - 1) // 2) % R], [M_FIXED])
    st = _i32(g)
    api._check(lib.blsgpu_multi_verify(1, api.POP, P(d_pks), n, P(d_sig), api._ptr(M_FIXED), len(M_FIXED), api.FMT_RAW_PROJ, P(st)))
    assert int(st.item()) == api.OK
    api._check(lib.blsgpu_multi_verify(1, api.POP, P(d_pks), n - 1, P(d_sig), api._ptr(M_FIXED), len(M_FIXED), api.FMT_RAW_PROJ, P(st)))
    assert int(st.item()) == api.INVALID_SIGNATURE          # one key short


def test_config4_aggregate_verify_262144_pairs(api, gpu):
    """AggregateSignature::verify (Basic: exercises the duplicate-message rule) over 262,144 (pk, msg) pairs; the
    aggregate is the device sum of the device signatures."""
    g, n = gpu, 262144
    lib, P, torch = g['lib'], g['P'], g['torch']
    msgs = [hashlib.sha256(SEED + i.to_bytes(8, 'little')).digest() for i in range(n)]
    d_pks, d_sigs, d_msgs, d_offs = g['sign'](1, api.BASIC, [(S0 + i) % R or 1 for i in range(n)], msgs)
    d_agg = torch.empty(144, dtype=torch.uint8, device=g['dev'])
    api._check(lib.blsgpu_sum_g1(P(d_sigs), n, api.FMT_RAW_PROJ, P(d_agg)))
    st = _i32(g)
    aux = torch.zeros(2, dtype=torch.int64, device=g['dev'])
    api._check(lib.blsgpu_aggregate_verify(1, api.BASIC, P(d_pks), P(d_msgs), P(d_offs), n, P(d_agg), api.FMT_RAW_PROJ, P(st), P(aux)))
    assert int(st.item()) == api.OK
    d_msgs[32 * 200001 + 5] ^= 1                              # one bit of one message
    api._check(lib.blsgpu_aggregate_verify(1, api.BASIC, P(d_pks), P(d_msgs), P(d_offs), n, P(d_agg), api.FMT_RAW_PROJ, P(st), P(aux)))
    assert int(st.item()) == api.INVALID_SIGNATURE
    d_msgs[32 * 200001:32 * 200002] = d_msgs[32 * 7:32 * 8]   # ... and a duplicate far apart
    api._check(lib.blsgpu_aggregate_verify(1, api.BASIC, P(d_pks), P(d_msgs), P(d_offs), n, P(d_agg), api.FMT_RAW_PROJ, P(st), P(aux)))
    assert (int(st.item()), aux.tolist()) == (api.DUPLICATE_MESSAGE, [7, 200001])


@pytest.mark.parametrize('sg', [1, 2])
def test_aggregate_verify_around_the_chunk_boundaries(api, gpu, sg):
    """The per-entry product form of the pairing product (blsgpu.hip run_miller_product_tree) at the sizes where its plan changes:
    n + 1 pairs = one machine round of lane pairs exactly (65,535 + the signature's), a round and a leftover chunk on the tail
    stream (65,536 + 1; 66,000 + 1: 465 leftovers), a round and a second chunk (70,000 + 1).  Each size: the valid aggregate,
    a tampered message in the LAST chunk and in the first, an identity key in the leftover (its pair enters as the line 1, its
    index comes back).  Both signature groups."""
    g = gpu
    lib, P, torch = g['lib'], g['P'], g['torch']
    nmax = 70000
    msgs = [hashlib.sha256(SEED + b'chunk' + i.to_bytes(8, 'little')).digest() for i in range(nmax)]
    d_pks, d_sigs, d_msgs, d_offs = g['sign'](sg, api.POP, [(S0 + 3 * i) % R or 1 for i in range(nmax)], msgs)
    pk_sz, sig_sz = (288, 144) if sg == 1 else (144, 288)
    st = _i32(g)
    aux = torch.zeros(2, dtype=torch.int64, device=g['dev'])
    d_agg = torch.empty(sig_sz, dtype=torch.uint8, device=g['dev'])
    for n in (65535, 65536, 66000, 70000):
        api._check((lib.blsgpu_sum_g1 if sg == 1 else lib.blsgpu_sum_g2)(P(d_sigs), n, api.FMT_RAW_PROJ, P(d_agg)))
        api._check(lib.blsgpu_aggregate_verify(sg, api.POP, P(d_pks), P(d_msgs), P(d_offs), n, P(d_agg), api.FMT_RAW_PROJ, P(st), P(aux)))
        assert int(st.item()) == api.OK, n
        for j in (n - 1, 5):
            d_msgs[32 * j + 3] ^= 1
            api._check(lib.blsgpu_aggregate_verify(sg, api.POP, P(d_pks), P(d_msgs), P(d_offs), n, P(d_agg), api.FMT_RAW_PROJ, P(st), P(aux)))
            assert int(st.item()) == api.INVALID_SIGNATURE, (n, j)
            d_msgs[32 * j + 3] ^= 1
        keep = d_pks[pk_sz * (n - 1):pk_sz * n].clone()
        d_pks[pk_sz * (n - 1):pk_sz * n] = 0                       # RAW_PROJ identity: Z = 0
        api._check(lib.blsgpu_aggregate_verify(sg, api.POP, P(d_pks), P(d_msgs), P(d_offs), n, P(d_agg), api.FMT_RAW_PROJ, P(st), P(aux)))
        assert int(st.item()) == api.PK_IDENTITY and int(aux[0].item()) == n, n      # 1-based index of the first identity key
        d_pks[pk_sz * (n - 1):pk_sz * n] = keep


@pytest.mark.parametrize('sg,mode', [(1, 0), (2, 0), (2, 1)])
def test_config5_verify_secure_65536_keys(api, gpu, sg, mode):
    """verify_secure over 65,536 keys: the library's own coefficients t_i give the valid secure aggregate
    (sum_i t_i sk_i) H(m) as one device signature; a different coefficient vector must fail."""
    g, n = gpu, 65536
    lib, P, torch = g['lib'], g['P'], g['torch']
    sks = [(S0 + i) % R or 1 for i in range(n)]
    d_pks, _, _, _ = g['sign'](sg, api.BASIC, sks, [b''] * n)
    pk_group, width = (2, 96) if sg == 1 else (1, 48)
    d_bytes = torch.empty(n * width, dtype=torch.uint8, device=g['dev'])
    api._check(lib.blsgpu_serialize(pk_group, P(d_pks), n, api.FMT_RAW_PROJ, api.FMT_LEGACY if mode else api.FMT_COMPRESSED, P(d_bytes), None))
    perm = torch.zeros(n, dtype=torch.int32, device=g['dev'])
    scal = torch.zeros(32 * n, dtype=torch.uint8, device=g['dev'])
    st = _i32(g)
    api._check(lib.blsgpu_secure_coefficients(P(d_bytes), n, width, P(perm), P(scal), P(st)))
    assert int(st.item()) == api.OK
    perm_h, scal_h = perm.cpu().tolist(), scal.cpu().numpy().tobytes()
    total = sum(int.from_bytes(scal_h[32 * p:32 * p + 32], 'little') * sks[perm_h[p]] for p in range(n)) % R
    _, d_sig, _, _ = g['sign'](sg, api.BASIC, [total], [M_FIXED])
    api._check(lib.blsgpu_verify_secure(sg, api.BASIC, P(d_pks), n, P(d_sig), api._ptr(M_FIXED), len(M_FIXED), mode, api.FMT_RAW_PROJ, P(st)))
    assert int(st.item()) == api.OK
    _, d_bad, _, _ = g['sign'](sg, api.BASIC, [(total + sks[12345]) % R], [M_FIXED])
    api._check(lib.blsgpu_verify_secure(sg, api.BASIC, P(d_pks), n, P(d_bad), api._ptr(M_FIXED), len(M_FIXED), mode, api.FMT_RAW_PROJ, P(st)))
    assert int(st.item()) == api.INVALID_SIGNATURE


def test_config2_verify_batch_65536(api, gpu):
    """configs[1] at its full size: 65,536 independent Signature<Bls12381G1Impl>::verify items signed on the device, 1 % with a
    flipped message bit, a few with swapped signatures / identity members: the exact verdict vector, and the first 4,096
    items item by item against the C oracle (oracle/c)."""
    torch, lib, P = gpu['torch'], gpu['lib'], gpu['P']
    n = 65536
    msgs = [hashlib.sha256(SEED + i.to_bytes(8, 'little')).digest() for i in range(n)]
    d_pks, d_sigs, d_msgs, d_offs = gpu['sign'](1, api.POP, [(S0 + i) % R for i in range(n)], msgs)
    bad = torch.arange(37, n, 100, device=gpu['dev'])
    d_msgs[bad * 32] ^= 1
    expect = torch.zeros(n, dtype=torch.int32, device=gpu['dev'])
    expect[bad] = api.INVALID_SIGNATURE
    sig_rows = d_sigs.view(n, 144)
    keep = sig_rows[[10, 11]].clone()
    sig_rows[10], sig_rows[11] = keep[1], keep[0]                 # swapped signatures: both items fail
    expect[10] = expect[11] = api.INVALID_SIGNATURE
    sig_rows[20, 96:] = 0                                          # Z = 0: the identity signature
    expect[20] = api.SIG_IDENTITY
    d_pks.view(n, 288)[21, 192:] = 0                               # identity public key
    expect[21] = api.PK_IDENTITY
    d_pks.view(n, 288)[22, 192:] = 0                               # both: the signature check comes first
    sig_rows[22, 96:] = 0
    expect[22] = api.SIG_IDENTITY
    d_st = torch.full((n,), -7, dtype=torch.int32, device=gpu['dev'])
    api._check(lib.blsgpu_verify_batch(1, api.POP, P(d_pks), P(d_sigs), P(d_msgs), P(d_offs), n, api.FMT_RAW_PROJ, P(d_st)))
    assert torch.equal(d_st, expect)
    sample = 4096
    bo = util.load_c_oracle()
    pks_h, sigs_h = d_pks[:sample * 288].cpu().numpy().tobytes(), d_sigs[:sample * 144].cpu().numpy().tobytes()
    blob = d_msgs[:sample * 32].cpu().numpy().tobytes()
    offs = (ctypes.c_uint64 * (sample + 1))(*[32 * i for i in range(sample + 1)])
    st = (ctypes.c_int32 * sample)()
    V = lambda x: ctypes.cast(x, ctypes.c_void_p)  # noqa: E731
    bo.bo_verify_batch(1, 2, V(ctypes.c_char_p(pks_h)), V(ctypes.c_char_p(sigs_h)), V(ctypes.c_char_p(blob)), V(offs), sample, V(st), 16)
    assert list(st) == d_st[:sample].cpu().tolist()


@pytest.mark.parametrize('sg', [1, 2])
def test_verify_batch_beyond_one_chunk(api, gpu, sg):
    """More items than one pass of the two-kernel Miller loop takes (65,536): a second, ragged chunk (70,001 items; Bls12381G2Impl: the
    two-pass line kernel), tampered items on both sides of the chunk boundary and in the last workgroup, identities in the second chunk:
    the exact verdict vector."""
    torch, lib, P = gpu['torch'], gpu['lib'], gpu['P']
    n = 70001
    pksz, sgsz = (288, 144) if sg == 1 else (144, 288)
    msgs = [hashlib.sha256(SEED + b'chunks' + i.to_bytes(8, 'little')).digest() for i in range(n)]
    d_pks, d_sigs, d_msgs, d_offs = gpu['sign'](sg, api.POP, [(S0 + 77 + i) % R for i in range(n)], msgs)
    bad = torch.tensor(sorted(set(list(range(11, n, 911)) + [65535, 65536, 65537, 69999, 70000])), device=gpu['dev'])
    d_msgs[bad * 32] ^= 1
    expect = torch.zeros(n, dtype=torch.int32, device=gpu['dev'])
    expect[bad] = api.INVALID_SIGNATURE
    d_sigs.view(n, sgsz)[66000, 2 * sgsz // 3:] = 0                 # Z = 0: identity signature in the second chunk
    expect[66000] = api.SIG_IDENTITY
    d_pks.view(n, pksz)[66001, 2 * pksz // 3:] = 0                  # identity key
    expect[66001] = api.PK_IDENTITY
    d_st = torch.full((n,), -7, dtype=torch.int32, device=gpu['dev'])
    api._check(lib.blsgpu_verify_batch(sg, api.POP, P(d_pks), P(d_sigs), P(d_msgs), P(d_offs), n, api.FMT_RAW_PROJ, P(d_st)))
    diff = torch.nonzero(d_st != expect).view(-1)
    assert diff.numel() == 0, diff[:10].tolist()
