"""BASELINE.json configs 1 and 3-5 beside their CPU legs (parity-test cases, not the bench line).

GPU side: config 1 (1,024 sequential single-item Signature::verify calls through the C ABI, the latency number) is timed
here; configs 3-5 are timed by `python bench.py --config 3|4|5` (one code path for 1 and N GPUs) and only quoted here when
--gpu-configs is given.
CPU side (SURVEY 8d "timing the reference's CPU path beside it"; the reference itself cannot be built here): the plain-C
oracle (oracle/c, kind "port", NOT blst) running the reference's own loops -- serial `g += key` (src/traits/pk_multi.rs:7-13),
one (n + 1)-pair Miller product (src/traits/sig_core.rs:149-178), n serial scalar multiplications
(src/secure_aggregation.rs:201-204) -- on 1 thread and on all host cores, on BOUNDED samples of the same device-signed
inputs (sizes stated per line; rates are per item, so they extrapolate linearly except for the O(n log n) key sort).
Prints one JSON object per line.   usage: python tools/bench_configs.py [--cpu-seconds 10] [--skip-gpu]"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

R = bench.R_ORDER


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cpu-seconds', type=float, default=8.0, help='target wall time per CPU leg (sample sizes are derived from a probe)')
    ap.add_argument('--skip-gpu', action='store_true')
    ap.add_argument('--skip-cpu', action='store_true')
    args = ap.parse_args()
    pkg = ge.import_pkg()
    api = pkg.api
    lib = api.init(0)
    dev = torch.device('cuda', 0)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    V = lambda b: ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p)  # noqa: E731
    msg = bench.FIXED_MSG
    cores = bench.host_cores()

    def emit(d):
        print(json.dumps(d), flush=True)

    def sign(sg, scheme, n, blob, msg_len):
        pksz, sgsz = (288, 144) if sg == 1 else (144, 288)
        d_msgs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        d_offs = (torch.arange(n + 1, dtype=torch.int64) * msg_len).to(dev)
        d_pks = torch.empty(n * pksz, dtype=torch.uint8, device=dev)
        d_sigs = torch.empty(n * sgsz, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        api._check(lib.blsgpu_sign_batch(sg, scheme, api._ptr(bench.sk_bytes(n, 0)), P(d_msgs), P(d_offs), n, P(d_pks), P(d_sigs)))
        return d_pks, d_sigs, d_msgs, d_offs

    # ---- config 1 on the GPU: 1,024 sequential single-item Signature::verify calls (host pointers, one item per call)
    n1 = 1024
    pks1, sigs1 = api.sign_batch(1, api.POP, [(bench.S0 + i) % R or 1 for i in range(n1)], [msg] * n1)
    if not args.skip_gpu:
        offs1 = (ctypes.c_uint64 * 2)(0, len(msg))
        st1 = ctypes.c_int32(-9)
        lat = []
        for i in range(n1):
            t = time.perf_counter()
            api._check(lib.blsgpu_verify_batch(1, api.POP, api._ptr(pks1[i]), api._ptr(sigs1[i]), api._ptr(msg), ctypes.cast(offs1, ctypes.c_void_p), 1, 0,
                                               ctypes.cast(ctypes.byref(st1), ctypes.c_void_p)))
            lat.append(time.perf_counter() - t)
            assert st1.value == 0
        lat.sort()
        emit({'config': 1, 'side': 'gpu', 'n': n1, 'calls': 'sequential, one item each, host pointers', 'mean_ms': 1e3 * sum(lat) / n1,
              'p50_ms': 1e3 * lat[n1 // 2], 'p99_ms': 1e3 * lat[min(n1 - 1, int(n1 * 0.99))], 'verifications_per_s': n1 / sum(lat)})
    if args.skip_cpu:
        return
    import util
    bo = util.load_c_oracle()
    cpu = {'kind': 'port', 'what': 'oracle/c plain-C restatement (not blst) of the reference loop', 'host_cores': cores}
    # ---- config 1 on the CPU: the same 1,024 calls, one thread (the reference's configs[0] is exactly this)
    lat = []
    for i in range(n1):
        t = time.perf_counter()
        r = bo.bo_verify(1, 2, pks1[i], sigs1[i], msg, len(msg))
        lat.append(time.perf_counter() - t)
        assert r == 0
    lat.sort()
    per_verify = sum(lat) / n1
    emit(dict(cpu, config=1, side='cpu', threads=1, n=n1, mean_ms=1e3 * per_verify, p50_ms=1e3 * lat[n1 // 2], verifications_per_s=1 / per_verify))
    pkb, sgb = b''.join(pks1), b''.join(sigs1)
    offs = (ctypes.c_uint64 * (n1 + 1))(*[32 * i for i in range(n1 + 1)])
    st = (ctypes.c_int32 * n1)()
    t = time.perf_counter()
    bo.bo_verify_batch(1, 2, V(pkb), V(sgb), V(msg * n1), ctypes.cast(offs, ctypes.c_void_p), n1, ctypes.cast(st, ctypes.c_void_p), cores)
    dt = time.perf_counter() - t
    assert not any(st)
    emit(dict(cpu, config=1, side='cpu', threads=cores, n=n1, seconds=dt, verifications_per_s=n1 / dt, note='the 1,024 items as one threaded batch'))

    # ---- config 3: serial G2 key sum + one verify
    n3 = 1 << 20
    d_pks, d_sigs, _, _ = sign(1, api.POP, n3, msg * n3, 32)
    agg = torch.empty(144, dtype=torch.uint8, device=dev)
    api._check(lib.blsgpu_sum_g1(P(d_sigs), n3, 0, P(agg)))
    pk_h, agg_h = d_pks.cpu().numpy().tobytes(), agg.cpu().numpy().tobytes()
    del d_pks, d_sigs
    probe = 4096
    t = time.perf_counter()
    bo.bo_multi_verify(1, 2, V(pk_h), probe, V(agg_h), msg, 32, 1)
    per_key = max((time.perf_counter() - t - 2 * per_verify) / probe, 1e-7)
    for th in (1, cores):
        ns = int(min(n3, max(probe, args.cpu_seconds * th / per_key)))
        t = time.perf_counter()
        r = bo.bo_multi_verify(1, 2, V(pk_h), ns, V(agg_h), msg, 32, th)
        dt = time.perf_counter() - t
        assert r == (0 if ns == n3 else 1)                 # a partial key set must not verify the full aggregate
        emit(dict(cpu, config=3, side='cpu', threads=th, sample_keys=ns, of=n3, seconds=dt, keys_per_s=ns / dt,
                  extrapolated_seconds_full=dt * n3 / ns))
    del pk_h

    # ---- config 4: n hash-to-curve + (n + 1)-pair Miller product + one final exponentiation
    n4 = 16384
    blob = b''.join(hashlib.sha256(bench.SEED + i.to_bytes(8, 'little')).digest() for i in range(n4))
    d_pks, d_sigs, _, _ = sign(1, api.BASIC, n4, blob, 32)
    pk_h, sg_h = d_pks.cpu().numpy().tobytes(), d_sigs.cpu().numpy().tobytes()
    offs4 = (ctypes.c_uint64 * (n4 + 1))(*[32 * i for i in range(n4 + 1)])
    aux = (ctypes.c_uint64 * 2)()
    per_pair = 1.0 * per_verify                            # hash + one Miller pair is about one verify without the final exp
    for th in (1, cores):
        ns = int(min(n4, max(64, args.cpu_seconds * th / per_pair)))
        api._check(lib.blsgpu_sum_g1(P(d_sigs), ns, 0, P(agg)))          # the aggregate of exactly the sampled pairs
        t = time.perf_counter()
        r = bo.bo_aggregate_verify(1, 0, V(pk_h), V(blob), ctypes.cast(offs4, ctypes.c_void_p), ns, V(agg.cpu().numpy().tobytes()), th,
                                   ctypes.cast(aux, ctypes.c_void_p))
        dt = time.perf_counter() - t
        assert r == 0
        emit(dict(cpu, config=4, side='cpu', threads=th, sample_pairs=ns, of=262144, seconds=dt, pairs_per_s=ns / dt,
                  extrapolated_seconds_full=dt * 262144 / ns))
    del d_pks, d_sigs

    # ---- config 5: sort + coefficient hashes + n serial 255-bit scalar multiplications + one verify
    n5 = 8192
    for sg, mode, name in ((1, 0, 'G1Impl/Modern'), (2, 0, 'G2Impl/Modern'), (2, 1, 'G2Impl/Legacy')):
        d_pks, d_sigs, _, _ = sign(sg, api.BASIC, n5, msg * n5, 32)
        pk_h = d_pks.cpu().numpy().tobytes()
        sgsz = 144 if sg == 1 else 288
        per_key = (3.0 if sg == 1 else 1.0) * 0.6 * per_verify        # a 255-bit G2 (G1) scalar multiplication, rough probe
        for th in (1, cores):
            ns = int(min(n5, max(64, args.cpu_seconds * th / per_key)))
            # the signature verify_secure accepts for exactly the sampled keys (the device's sign-side twin, tests/ check it)
            st5, aggs = api.aggregate_secure(sg, [pk_h[(432 - sgsz) * i:(432 - sgsz) * (i + 1)] for i in range(ns)],
                                             [d_sigs[sgsz * i:sgsz * (i + 1)].cpu().numpy().tobytes() for i in range(ns)], mode)
            assert st5 == 0
            t = time.perf_counter()
            r = bo.bo_verify_secure_mt(sg, 0, V(pk_h), ns, V(aggs), msg, 32, mode, th)
            dt = time.perf_counter() - t
            assert r == 0
            emit(dict(cpu, config=5, side='cpu', variant=name, threads=th, sample_keys=ns, of=65536, seconds=dt, keys_per_s=ns / dt,
                      extrapolated_seconds_full=dt * 65536 / ns))
        del d_pks, d_sigs


if __name__ == '__main__':
    main()
