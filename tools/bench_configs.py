"""Wall-clock of BASELINE.json configs 1 and 3-5 on one MI355X (parity-test cases, not the bench line): inputs are made on
the device with blsgpu_sign_batch and stay resident in HBM.  Prints one JSON object per config.
usage: python tools/bench_configs.py [--scale 1.0]"""
import argparse, ctypes, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import __graft_entry__ as ge

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--scale', type=float, default=1.0); args = ap.parse_args()
    pkg = ge.import_pkg(); api = pkg.api; lib = api.init(0)
    dev = torch.device('cuda', 0)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    seed = hashlib.sha256(b'blsgpu-bench-v1').digest(); s0 = int.from_bytes(seed, 'big') % R

    def make(sg, scheme, n, msgs=None, one_msg=None):
        sks = [(s0 + i) % R or 1 for i in range(n)]
        skb = b''.join(s.to_bytes(32, 'little') for s in sks)
        if one_msg is not None:
            blob = one_msg * n; offs = torch.arange(n + 1, dtype=torch.int64) * len(one_msg)
        else:
            blob = b''.join(msgs); offs = torch.arange(n + 1, dtype=torch.int64) * 32
        d_msgs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev); d_offs = offs.to(dev)
        pksz, sgsz = (288, 144) if sg == 1 else (144, 288)
        d_pks = torch.empty(n * pksz, dtype=torch.uint8, device=dev); d_sigs = torch.empty(n * sgsz, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        api._check(lib.blsgpu_sign_batch(sg, scheme, api._ptr(skb), P(d_msgs), P(d_offs), n, P(d_pks), P(d_sigs)))
        return sks, d_pks, d_sigs, d_msgs, d_offs

    def timed(fn, reps=3):
        fn(); ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        return min(ts)

    st = ctypes.c_int32(-9)
    # ---- config 1: 1,024 sequential single-item Signature::verify calls (the reference's own CPU-runnable case): the
    # latency / plumbing number -- each call is one blsgpu_verify_batch of one item with host pointers, as a caller that
    # keeps the reference's one-signature-per-call API would issue it
    n1 = max(16, int(1024 * args.scale)); msg = hashlib.sha256(seed + b'fixed').digest()
    sks1 = [(s0 + i) % R or 1 for i in range(n1)]
    pks1, sigs1 = api.sign_batch(1, api.POP, sks1, [msg] * n1)
    offs1 = (ctypes.c_uint64 * 2)(0, len(msg)); st1 = ctypes.c_int32(-9)
    lat = []
    for i in range(n1):
        t = time.perf_counter()
        api._check(lib.blsgpu_verify_batch(1, api.POP, api._ptr(pks1[i]), api._ptr(sigs1[i]), api._ptr(msg), ctypes.cast(offs1, ctypes.c_void_p), 1, 0,
                                           ctypes.cast(ctypes.byref(st1), ctypes.c_void_p)))
        lat.append(time.perf_counter() - t)
        assert st1.value == 0
    lat.sort()
    print(json.dumps({'config': 1, 'n': n1, 'calls': 'sequential, one item each, host pointers', 'mean_ms': 1e3 * sum(lat) / n1,
                      'p50_ms': 1e3 * lat[n1 // 2], 'p99_ms': 1e3 * lat[min(n1 - 1, int(n1 * 0.99))], 'verifications_per_s': n1 / sum(lat)}), flush=True)
    # ---- config 3: MultiSignature::verify, 1,048,576 G2 public keys, one message
    n = int(1048576 * args.scale); msg = hashlib.sha256(seed + b'fixed').digest()
    sks, d_pks, d_sigs, _, _ = make(1, api.POP, n, one_msg=msg)
    agg = torch.empty(144, dtype=torch.uint8, device=dev)
    api._check(lib.blsgpu_sum_g1(P(d_sigs), n, 0, P(agg)))
    api.profile_enable(True)
    t = timed(lambda: api._check(lib.blsgpu_multi_verify(1, api.POP, P(d_pks), n, P(agg), api._ptr(msg), len(msg), 0, ctypes.byref(st))))
    prof = api.profile_read(); api.profile_enable(False)
    print(json.dumps({'config': 3, 'n': n, 'status': st.value, 'seconds': t, 'keys_per_s': n / t, 'GBps_algorithmic': 288 * n / t / 1e9,
                      'kernel_ms': {k: round(v[0] / v[1], 3) for k, v in prof.items()}, 'launches': {k: v[1] for k, v in prof.items()}}), flush=True)
    del d_pks, d_sigs
    # ---- config 4: AggregateSignature::verify, 262,144 distinct (pk, msg), Basic
    n = int(262144 * args.scale); msgs = [hashlib.sha256(seed + i.to_bytes(8, 'little')).digest() for i in range(n)]
    sks, d_pks, d_sigs, d_msgs, d_offs = make(1, api.BASIC, n, msgs=msgs)
    api._check(lib.blsgpu_sum_g1(P(d_sigs), n, 0, P(agg)))
    aux = (ctypes.c_uint64 * 2)()
    api.profile_enable(True)
    t = timed(lambda: api._check(lib.blsgpu_aggregate_verify(1, api.BASIC, P(d_pks), P(d_msgs), P(d_offs), n, P(agg), 0, ctypes.byref(st), ctypes.cast(aux, ctypes.c_void_p))), reps=2)
    prof = api.profile_read(); api.profile_enable(False)
    print(json.dumps({'config': 4, 'n': n, 'status': st.value, 'seconds': t, 'pairs_per_s': n / t,
                      'kernel_ms': {k: round(v[0] / v[1], 3) for k, v in prof.items()}, 'launches': {k: v[1] for k, v in prof.items()}}), flush=True)
    del d_pks, d_sigs, d_msgs
    # ---- config 5: verify_secure, 65,536 keys: G1Impl Modern, G2Impl Modern, G2Impl Legacy
    n = int(65536 * args.scale)
    for sg, mode, name in ((1, 0, 'G1Impl/Modern'), (2, 0, 'G2Impl/Modern'), (2, 1, 'G2Impl/Legacy')):
        sks, d_pks, d_sigs, _, _ = make(sg, api.BASIC, n, one_msg=msg)
        pk_group = 2 if sg == 1 else 1
        kb = torch.empty(n * (96 if sg == 1 else 48), dtype=torch.uint8, device=dev)
        api._check(lib.blsgpu_serialize(pk_group, P(d_pks), n, 0, api.FMT_LEGACY if mode else api.FMT_COMPRESSED, P(kb), None))
        w = 96 if sg == 1 else 48
        kbl = kb.cpu().numpy().tobytes()
        stc, perm, ts = api.secure_coefficients([kbl[w * i:w * (i + 1)] for i in range(n)])
        idx = torch.tensor(perm, dtype=torch.int64, device=dev)
        sgsz = 144 if sg == 1 else 288
        sig_sorted = d_sigs.view(n, sgsz)[idx].contiguous().view(-1)
        scal = torch.frombuffer(bytearray(b''.join(t_.to_bytes(32, 'little') for t_ in ts)), dtype=torch.uint8).to(dev)
        aggs = torch.empty(sgsz, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        fn = lib.blsgpu_msm_g1 if sg == 1 else lib.blsgpu_msm_g2
        api._check(fn(P(sig_sorted), P(scal), n, 0, P(aggs)))
        api.profile_enable(True)
        t = timed(lambda: api._check(lib.blsgpu_verify_secure(sg, api.BASIC, P(d_pks), n, P(aggs), api._ptr(msg), len(msg), mode, 0, ctypes.byref(st))), reps=2)
        prof = api.profile_read(); api.profile_enable(False)
        print(json.dumps({'config': 5, 'variant': name, 'n': n, 'status': st.value, 'seconds': t, 'keys_per_s': n / t,
                          'kernel_ms': {k: round(v[0] / v[1], 3) for k, v in prof.items()}, 'launches': {k: v[1] for k, v in prof.items()}}), flush=True)


if __name__ == '__main__':
    main()
