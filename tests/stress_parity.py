"""Randomised differential campaign on the GPU box: the C ABI against oracle/c (test infrastructure, the checker) on inputs no
fixed test holds -- random batch sizes on both sides of every path boundary (1 / 128 / 512 / 1,024 / 4,096 items), random
message lengths, random tampering (swapped signatures, flipped message bytes, identity points), all three schemes, both
groups; hash-to-curve of random messages; MultiSignature::verify, verify_secure and AggregateSignature::verify of random key sets.  Not collected by pytest
(minutes, not seconds): python tests/stress_parity.py --seconds 480 [--seed S].  Prints one line per round and a summary; exits
non-zero at the first mismatch with the seed and the round that reproduce it."""
import argparse
import ctypes
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import util  # noqa: E402
from util import c  # noqa: E402

V = lambda x: ctypes.cast(x, ctypes.c_void_p)  # noqa: E731
SIZES = [1, 2, 3, 17, 64, 127, 128, 129, 255, 256, 257, 511, 512, 513, 700, 1023, 1024, 1025, 2000, 4096, 4097, 6145, 9000]


def oracle_statuses(bo, sg, scheme, pks, sigs, msgs):
    n = len(pks)
    offs, blob = [0], b''
    for m in msgs:
        blob += m
        offs.append(len(blob))
    o = (ctypes.c_uint64 * (n + 1))(*offs)
    st = (ctypes.c_int32 * n)()
    bo.bo_verify_batch(sg, scheme, V(ctypes.c_char_p(b''.join(pks))), V(ctypes.c_char_p(b''.join(sigs))), V(ctypes.c_char_p(blob or b'\0')), V(o), n, V(st), 16)
    return list(st)


def round_verify_batch(api, bo, rng, log):
    sg, scheme, n = rng.choice((1, 2)), rng.choice((0, 1, 2)), rng.choice(SIZES)
    sks = [rng.randrange(1, c.R) for _ in range(n)]
    msgs = [rng.randbytes(rng.choice((0, 1, 31, 32, 32, 32, 55, 56, 64, 100, 200))) for _ in range(n)]
    pks, sigs = api.sign_batch(sg, scheme, sks, msgs)
    pks, sigs, msgs = list(pks), list(sigs), list(msgs)
    zero_z = lambda b: b[:2 * len(b) // 3] + bytes(len(b) // 3)  # noqa: E731
    for _ in range(max(1, n // 20)):
        i, kind = rng.randrange(n), rng.randrange(5)
        if kind == 0 and n > 1:
            j = rng.randrange(n)
            sigs[i], sigs[j] = sigs[j], sigs[i]
        elif kind == 1:
            msgs[i] = msgs[i] + b'x'
        elif kind == 2 and msgs[i]:
            k = rng.randrange(len(msgs[i]))
            msgs[i] = msgs[i][:k] + bytes([msgs[i][k] ^ (1 << rng.randrange(8))]) + msgs[i][k + 1:]
        elif kind == 3:
            sigs[i] = zero_z(sigs[i])
        else:
            pks[i] = zero_z(pks[i])
    got = api.verify_batch(sg, scheme, pks, sigs, msgs)
    want = oracle_statuses(bo, sg, scheme, pks, sigs, msgs)
    log('verify_batch sg=%d scheme=%d n=%d bad=%d' % (sg, scheme, n, sum(1 for s in want if s)))
    return list(got) == want


def round_hash(api, bo, rng, log):
    group, n = rng.choice((1, 2)), rng.choice((1, 5, 128, 129, 400, 600, 3000))
    msgs = [rng.randbytes(rng.randrange(0, 300)) for _ in range(n)]
    dst = rng.choice((b'BLS_SIG_BLS12381G%d_XMD:SHA-256_SSWU_RO_NUL_' % group, b'QUUX-V01-CS02-with-BLS12381G%d_XMD:SHA-256_SSWU_RO_' % group, b'd' * rng.randrange(1, 60)))
    got = api.serialize(group, api.hash_to_point(group, msgs, dst))
    out = ctypes.create_string_buffer(48 * group)
    ok = True
    for i in rng.sample(range(n), min(n, 200)):
        bo.bo_hash_to_point(group, msgs[i], len(msgs[i]), dst, len(dst), out)
        ok = ok and got[i] == out.raw
    log('hash_to_point group=%d n=%d dst=%d bytes' % (group, n, len(dst)))
    return ok


def round_multi(api, bo, rng, log):
    sg, scheme, n = rng.choice((1, 2)), rng.choice((0, 2)), rng.choice((1, 2, 15, 16, 17, 300, 4096, 4097, 20000))
    secure = rng.random() < 0.5
    msg = rng.randbytes(rng.choice((0, 32, 77)))
    sks = [rng.randrange(1, c.R) for _ in range(n)]
    pks, sigs = api.sign_batch(sg, scheme, sks, [msg] * n)
    pks, sigs = list(pks), list(sigs)
    mode = rng.randrange(3)                      # 0: valid, 1: one signature replaced, 2: message changed
    if mode == 1:
        sigs[rng.randrange(n)] = api.sign_batch(sg, scheme, [rng.randrange(1, c.R)], [msg])[1][0]
    width = 288 if sg == 1 else 144
    swidth = 144 if sg == 1 else 288
    if secure:
        ser = rng.choice((0, 1)) if sg == 2 else 0
        st, agg = api.aggregate_secure(sg, pks, sigs, ser)
        assert st == 0
        vmsg = msg + (b'!' if mode == 2 else b'')
        got = api.verify_secure(sg, scheme, pks, agg, vmsg, ser)
        want = bo.bo_verify_secure_mt(sg, scheme, V(ctypes.c_char_p(b''.join(pks))), n, V(ctypes.c_char_p(agg)), vmsg, len(vmsg), ser, 16)
        log('verify_secure sg=%d scheme=%d n=%d ser=%d mode=%d -> %s' % (sg, scheme, n, ser, mode, want))
    else:
        agg = api.point_sum(2 if sg == 2 else 1, sigs)
        vmsg = msg + (b'!' if mode == 2 else b'')
        got = api.multi_verify(sg, scheme, pks, agg, vmsg)
        want = bo.bo_multi_verify(sg, scheme, V(ctypes.c_char_p(b''.join(pks))), n, V(ctypes.c_char_p(agg)), vmsg, len(vmsg), 16)
        log('multi_verify sg=%d scheme=%d n=%d mode=%d -> %s' % (sg, scheme, n, mode, want))
    assert len(pks[0]) == width and len(sigs[0]) == swidth
    return int(got) == int(want)


def round_aggregate(api, bo, rng, log):
    """AggregateSignature::verify: n (key, message) pairs under one summed signature; valid, a changed message, an identity key,
    a repeated message (an error for Basic, fine for the others), an identity signature"""
    sg, scheme, n = rng.choice((1, 2)), rng.choice((0, 1, 2)), rng.choice((1, 2, 3, 4, 50, 63, 64, 65, 67, 191, 192, 193, 1000, 1027, 5000, 16384, 16385, 20000))   # 64: the per-entry product form starts; 16,384 / 16,385 pairs proper: four / two lanes per item in the line kernel
    sks = [rng.randrange(1, c.R) for _ in range(n)]
    msgs = [i.to_bytes(4, 'big') + rng.randbytes(rng.choice((0, 28, 60))) for i in range(n)]
    pks, sigs = api.sign_batch(sg, scheme, sks, msgs)
    pks, msgs = list(pks), list(msgs)
    agg = api.point_sum(sg, list(sigs))
    mode = rng.randrange(5)
    if mode == 1:
        msgs[rng.randrange(n)] += b'?'
    elif mode == 2:
        i = rng.randrange(n)
        pks[i] = pks[i][:2 * len(pks[i]) // 3] + bytes(len(pks[i]) // 3)
    elif mode == 3 and n > 1:
        i, j = rng.sample(range(n), 2)
        msgs[i] = msgs[j]
    elif mode == 4:
        agg = agg[:2 * len(agg) // 3] + bytes(len(agg) // 3)
    got = api.aggregate_verify(sg, scheme, pks, msgs, agg)
    offs, blob = [0], b''
    for m in msgs:
        blob += m
        offs.append(len(blob))
    o = (ctypes.c_uint64 * (n + 1))(*offs)
    aux = (ctypes.c_uint64 * 2)()
    st = bo.bo_aggregate_verify(sg, scheme, V(ctypes.c_char_p(b''.join(pks))), V(ctypes.c_char_p(blob)), V(o), n, V(ctypes.c_char_p(agg)), 16, V(aux))
    log('aggregate_verify sg=%d scheme=%d n=%d mode=%d -> %s' % (sg, scheme, n, mode, st))
    return got == (st, (aux[0], aux[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=480.0)
    ap.add_argument('--seed', type=int, default=int(time.time()))
    args = ap.parse_args()
    import __graft_entry__ as ge
    api = ge.import_pkg().api
    api.init()
    bo = util.load_c_oracle()
    rng = random.Random(args.seed)
    t0, rounds, counts = time.time(), 0, {}
    print('seed', args.seed, flush=True)
    while time.time() - t0 < args.seconds:
        kind = rng.choices((round_verify_batch, round_hash, round_multi, round_aggregate), (5, 2, 2, 3))[0]
        state = rng.getstate()
        line = []
        ok = kind(api, bo, rng, line.append)
        rounds += 1
        counts[kind.__name__] = counts.get(kind.__name__, 0) + 1
        print('%5d %6.1fs %s %s' % (rounds, time.time() - t0, 'ok ' if ok else 'MISMATCH', ' '.join(line)), flush=True)
        if not ok:
            print('FAILED: seed %d round %d (rng state hash %d)' % (args.seed, rounds, hash(state) & 0xffffffff))
            sys.exit(1)
    print('all %d rounds agree with the C restatement: %s' % (rounds, counts))


if __name__ == '__main__':
    main()
