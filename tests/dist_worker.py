"""Worker for the world_size-2 tests of agora-blsful_amd/dist.py (spawned by tests/test_dist.py).
argv: rank world port backend('fake'|'hip') outfile"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    rank, world, port, backend, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import util
    from util import c, ref
    import __graft_entry__ as ge
    pkg = ge.import_pkg()
    from agora_blsful_amd import dist as bd
    import torch
    if backend == 'fake':
        import fake_backend
        ops = fake_backend.FakeOps()
    else:
        ops = pkg.api.TensorOps(torch.device('cuda', 0))      # both ranks share the one card of the GPU box
    sh = bd.Sharded(ops, dist)
    dev = ops.device

    def T(rows):
        """list of byte strings -> one uint8 tensor on the rank's device"""
        b = b''.join(rows)
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else torch.zeros(0, dtype=torch.uint8, device=dev)

    def M(msgs):
        """messages -> (blob tensor, int64 offsets tensor)"""
        offs, t = [0], 0
        for m in msgs:
            t += len(m)
            offs.append(t)
        return T(msgs), torch.tensor(offs, dtype=torch.int64, device=dev)

    def L(t):
        return [int(x) for x in t.cpu().tolist()]

    res = {}
    rng = random.Random(77)                      # same data on every rank; each rank uses its shard
    for C, sg in ((ref.G1Impl, 1), (ref.G2Impl, 2)):
        pkraw, sigraw = (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)
        n = 7
        sks = [ref.keygen_from_hash(bytes([i, sg]) * 16) for i in range(n)]
        pks = [ref.public_key(C, s) for s in sks]
        lo, hi = bd.shard_range(n, rank, world)
        nl = hi - lo
        # config 2
        msgs = [b'item %d' % i for i in range(n)]
        sigs = [ref.sign(C, ref.POP, s, m) for s, m in zip(sks, msgs)]
        msgs_t = list(msgs)
        msgs_t[5] = b'tampered'
        praw, sraw = [pkraw(p, rng) for p in pks], [sigraw(s, rng) for s in sigs]
        res['verify_batch_%d' % sg] = L(sh.verify_batch(sg, ref.POP, T(praw[lo:hi]), T(sraw[lo:hi]), *M(msgs_t[lo:hi]), nl, gather=True))
        # config 3
        m1 = b'one message'
        msig = ref.aggregate_signatures(C, [ref.sign(C, ref.POP, s, m1) for s in sks])
        res['multi_ok_%d' % sg] = sh.multi_verify(sg, ref.POP, T(praw[lo:hi]), nl, T([sigraw(msig, rng)]), m1)
        res['multi_bad_%d' % sg] = sh.multi_verify(sg, ref.POP, T(praw[lo:hi]), nl, T([sigraw(msig, rng)]), b'other')
        # config 4
        for scheme in (ref.BASIC, ref.AUG):
            asig = ref.aggregate_signatures(C, [ref.sign(C, scheme, s, m) for s, m in zip(sks, msgs)])

            def agg(pk_rows, msg_rows, sig_pt):
                st, aux = sh.aggregate_verify(sg, scheme, T(pk_rows[lo:hi]), *M(msg_rows[lo:hi]), nl, T([sigraw(sig_pt, rng) if sig_pt is not None else sigraw(None)]), lo)
                return [st, [int(aux[0]), int(aux[1])]]
            res['agg_ok_%d_%d' % (sg, scheme)] = agg(praw, msgs, asig)
            res['agg_bad_%d_%d' % (sg, scheme)] = agg(praw, msgs_t, asig)
            dup = list(msgs)
            dup[6] = dup[1]                        # duplicate across the shard boundary
            res['agg_dup_%d_%d' % (sg, scheme)] = agg(praw, dup, asig)
            pid = list(praw)
            pid[4] = pkraw(None)
            pid[5] = pkraw(None)
            res['agg_pkid_%d_%d' % (sg, scheme)] = agg(pid, msgs, asig)
            res['agg_sigid_%d_%d' % (sg, scheme)] = agg(pid, msgs, None)
        # config 5
        modes = [0] if sg == 1 else [0, 1]
        for mode in modes:
            ssigs = [C.sig_curve.mul(C.hash_to_point(m1, C.DST[ref.AUG]), s) for s in sks]
            agg_pt = ref.aggregate_secure(C, pks, ssigs, None if sg == 1 else mode)
            res['secure_ok_%d_%d' % (sg, mode)] = sh.verify_secure(sg, ref.AUG, T(praw[lo:hi]), nl, T([sigraw(agg_pt, rng)]), m1, lo, mode)
            sub = praw[lo:hi][:-1] if rank == world - 1 else praw[lo:hi]     # one key missing: the aggregate no longer verifies
            res['secure_sub_%d_%d' % (sg, mode)] = sh.verify_secure(sg, ref.AUG, T(sub), len(sub), T([sigraw(agg_pt, rng)]), m1, lo, mode)
        # N3: proofs of possession, one forged
        pops = [ref.pop_prove(C, s) for s in sks]
        pops[2] = pops[3]
        popraw = [sigraw(q, rng) for q in pops]
        res['pop_%d' % sg] = L(sh.pop_verify_batch(sg, T(praw[lo:hi]), T(popraw[lo:hi]), nl, gather=True))
        # N4: proofs of knowledge (one with a wrong challenge, one with an identity commitment) and signcryption validity
        us, vs, ys = [], [], []
        for i, (s_, m) in enumerate(zip(sks, msgs)):
            x, y = 1000 + i, 77 + i
            u, v = ref.sig_proof_generate(C, ref.sign(C, ref.BASIC, s_, m), m, C.DST[ref.BASIC], x, y)
            us.append(u), vs.append(v), ys.append(y)
        ys[1] += 1
        us[6] = None
        uraw, vraw = [sigraw(q, rng) for q in us], [sigraw(q, rng) for q in vs]
        yraw = [y.to_bytes(32, 'little') for y in ys]
        res['proof_%d' % sg] = L(sh.sig_proof_verify_batch(sg, ref.BASIC, T(uraw[lo:hi]), T(vraw[lo:hi]), T(praw[lo:hi]), T(yraw[lo:hi]), *M(msgs[lo:hi]), nl, gather=True))
        cu = [C.pk_curve.mul(C.pk_gen, 31 + i) for i in range(n)]
        cv = [b'ciphertext body %d' % i for i in range(n)]
        cw = [C.sig_curve.mul(ref.signcrypt_compute_w(C, cu[i], cv[i], C.DST[ref.POP]), 31 + i) for i in range(n)]
        cv[4] = b'tampered body'
        st = L(sh.signcrypt_valid_batch(sg, ref.POP, T([pkraw(q, rng) for q in cu][lo:hi]), T([sigraw(q, rng) for q in cw][lo:hi]), *M(cv[lo:hi]), nl, gather=True))
        res['signcrypt_%d' % sg] = [x == 0 for x in st]
        # N1: sign-side secure aggregation, with a duplicated key across the shard boundary (first occurrence wins)
        for mode in modes:
            dk, ds = list(pks), list(ssigs)
            dk[5], ds[5] = dk[1], ssigs[5]
            want = ref.aggregate_secure(C, dk, ds, None if sg == 1 else mode)
            dkraw, dsraw = [pkraw(p, rng) for p in dk], [sigraw(s, rng) for s in ds]
            st, agg_raw = sh.aggregate_secure(sg, T(dkraw[lo:hi]), T(dsraw[lo:hi]), nl, lo, mode)
            got = bytes(ops.serialize(sg, agg_raw, 1).cpu().numpy().tobytes())
            res['aggsec_%d_%d' % (sg, mode)] = [st, got.hex() == (c.g1_compress(want) if sg == 1 else c.g2_compress(want)).hex()]
        empty = torch.zeros(0, dtype=torch.uint8, device=dev)
        res['secure_empty_%d' % sg] = [sh.verify_secure(sg, ref.BASIC, empty, 0, T([sigraw(None)]), m1, 0), sh.verify_secure(sg, ref.BASIC, empty, 0, T([sigraw(msig, rng)]), m1, 0)]
        # an EMPTY shard (world 2, one pair): the neutral record of aggregate_partial is folded with the other rank's
        one_sig = ref.sign(C, ref.BASIC, sks[0], msgs[0])
        l1, h1 = bd.shard_range(1, rank, world)
        st, aux = sh.aggregate_verify(sg, ref.BASIC, T(praw[l1:h1]), *M(msgs[l1:h1]), h1 - l1, T([sigraw(one_sig, rng)]), l1)
        res['agg_one_%d' % sg] = [st, [int(aux[0]), int(aux[1])]]
    json.dump(res, open(outfile, 'w'))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
