"""Worker for tests/test_gpu_round2.py::test_in_library_multi_device: the four verify entry points through
blsgpu_init_devices (one process driving several devices; BLSGPU_FAKE_DEVICES maps the logical devices onto the one GPU
of the test box) must return exactly what the single-device library returns, for host and for device-resident inputs."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import torch
    import __graft_entry__ as ge
    pkg = ge.import_pkg()
    api = pkg.api
    lib = api.load_library()
    n = 200
    import util
    res = {}

    def run_all(tag):
        out = {}
        for sg in (1, 2):
            sks = [77 + 3 * i for i in range(n)]
            msgs = [b'message %d' % i for i in range(n)]
            pks, sigs = api.sign_batch(sg, api.BASIC, sks, msgs)
            # config 2: tampered items, identity members
            m2 = list(msgs)
            m2[5] = b'tampered'
            m2[150] = b'x'
            s2 = list(sigs)
            s2[60] = s2[61]
            out['vb_%d' % sg] = api.verify_batch(sg, api.BASIC, pks, s2, m2)
            # config 3
            pk1, sig1 = api.sign_batch(sg, api.POP, sks, [b'one'] * n)
            agg1 = api.point_sum(sg, sig1)
            out['mv_%d' % sg] = [api.multi_verify(sg, api.POP, pk1, agg1, b'one'), api.multi_verify(sg, api.POP, pk1, agg1, b'two'),
                                 api.multi_verify(sg, api.AUG, pk1, agg1, b'one')]
            # config 4: valid, tampered, duplicate, identity key, identity signature
            agg = api.point_sum(sg, sigs)
            dup = list(msgs)
            dup[180] = dup[7]
            pid = list(pks)
            pid[120] = pid[33] = (util.g2_raw(None) if sg == 1 else util.g1_raw(None))
            ident_sig = (util.g1_raw(None) if sg == 1 else util.g2_raw(None))
            out['av_%d' % sg] = [api.aggregate_verify(sg, api.BASIC, pks, msgs, agg), api.aggregate_verify(sg, api.BASIC, pks, m2, agg),
                                 api.aggregate_verify(sg, api.BASIC, pks, dup, agg), api.aggregate_verify(sg, api.AUG, pks, dup, agg),
                                 api.aggregate_verify(sg, api.BASIC, pid, msgs, agg), api.aggregate_verify(sg, api.BASIC, pid, msgs, ident_sig)]
            # config 5
            for mode in ([0] if sg == 1 else [0, 1]):
                st, sagg = api.aggregate_secure(sg, pk1, sig1, mode)
                out['vs_%d_%d' % (sg, mode)] = [st, api.verify_secure(sg, api.POP, pk1, sagg, b'one', mode), api.verify_secure(sg, api.POP, pk1, sagg, b'two', mode),
                                                api.verify_secure(sg, api.POP, pk1[:-1], sagg, b'one', mode)]
            # device-resident inputs through the same entry points
            dev = torch.device('cuda', 0)
            T = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)  # noqa: E731
            P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
            offs, t = [0], 0
            for m in m2:
                t += len(m)
                offs.append(t)
            d_pks, d_sigs, d_msgs, d_offs = T(b''.join(pks)), T(b''.join(s2)), T(b''.join(m2)), torch.tensor(offs, dtype=torch.int64, device=dev)
            d_st = torch.full((n,), -5, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            api._check(lib.blsgpu_verify_batch(sg, api.BASIC, P(d_pks), P(d_sigs), P(d_msgs), P(d_offs), n, 0, P(d_st)))
            out['vb_dev_%d' % sg] = d_st.cpu().tolist()
            d_agg, d_one = T(agg), torch.full((1,), -5, dtype=torch.int32, device=dev)
            d_aux = torch.zeros(2, dtype=torch.int64, device=dev)
            d_sigs_ok = T(b''.join(sigs))
            torch.cuda.synchronize()
            api._check(lib.blsgpu_aggregate_verify(sg, api.BASIC, P(d_pks), P(d_msgs), P(d_offs), n, P(d_agg), 0, P(d_one), P(d_aux)))
            out['av_dev_%d' % sg] = [int(d_one.item()), d_aux.cpu().tolist()]
            del d_sigs_ok
        res[tag] = out

    assert lib.blsgpu_init(0) == 0
    run_all('single')
    lib.blsgpu_shutdown()
    os.environ['BLSGPU_FAKE_DEVICES'] = '3'
    os.environ['BLSGPU_SHARD_MIN'] = '16'
    nd = lib.blsgpu_init_devices(0)
    assert nd == 3 and lib.blsgpu_device_count() == 3, nd
    run_all('multi')
    lib.blsgpu_shutdown()
    res['equal'] = res['single'] == res['multi']
    json.dump(res, open(sys.argv[1], 'w'))


if __name__ == '__main__':
    main()
