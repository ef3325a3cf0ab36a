#!/usr/bin/env python
"""Benchmark of the hot path: batch BLS12-381 signature verification on MI355X.

Workload (BASELINE.json configs[1]): N independent (pk, msg, sig) Signature<Bls12381G1Impl>::verify items per GPU
(default 65,536, 32-byte messages, ProofOfPossession scheme = the reference default), inputs already resident in HBM
in RAW_PROJ form, 1 % of the items tampered as negative controls.  A "step" = one pass over the batch through the
C ABI (blsgpu_verify_batch).  One process per GPU; N > 1 ranks shard independent batches (weak scaling, no
collective on the data path); timing = barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      the dominant kernel's algorithmic HBM bytes / its HIP-event-measured duration vs 8 TB/s
  cpu_baseline  the oracle timed on a bounded sample of the same workload on this box's host cores
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
SEED = hashlib.sha256(b'blsgpu-bench-v1').digest()
S0 = int.from_bytes(SEED, 'big') % R_ORDER
ALG_BYTES_PER_VERIFY = 468          # pk 288 + sig 144 + msg 32 + status 4
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FPMUL_PEAK_G = 66.6                 # profiles/ubench_r01_fp28.txt: best measured Fp-multiplication-equivalent rate of the 28-bit-limb
                                    # multiplier chip-wide (fused two-product pass, 2-4 waves/SIMD): the integer-VALU roofline


def gen_inputs(n, base):
    sks = [(S0 + base + i) % R_ORDER or 1 for i in range(n)]
    msgs = [hashlib.sha256(SEED + (base + i).to_bytes(8, 'little')).digest() for i in range(n)]
    return sks, msgs


def host_cores():
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(d_pks, d_sigs, msgs_host, sample, n_total):
    """The C oracle (oracle/c, kind "port") on the first `sample` items of the very batch the GPU verified (tampered
    ones included), on all host cores of this box: contiguous item ranges, one pthread per core."""
    import util
    bo = util.load_c_oracle()
    cores = host_cores()
    pks = d_pks[:sample * 288].cpu().numpy().tobytes()
    sigs = d_sigs[:sample * 144].cpu().numpy().tobytes()
    blob = b''.join(msgs_host[:sample])
    offs = (ctypes.c_uint64 * (sample + 1))(*[32 * i for i in range(sample + 1)])
    st = (ctypes.c_int32 * sample)()
    V = lambda x: ctypes.cast(x, ctypes.c_void_p)  # noqa: E731
    t0 = time.perf_counter()
    bo.bo_verify_batch(1, 2, V(ctypes.c_char_p(pks)), V(ctypes.c_char_p(sigs)), V(ctypes.c_char_p(blob)), V(offs), sample, V(st), cores)
    dt = time.perf_counter() - t0
    return {'value': sample / dt, 'unit': 'verifications/s', 'cores': cores, 'kind': 'port',
            'sample': 'first %d of the %d items of the GPU batch, oracle/c plain-C restatement (not blst), %d threads, %.1f s wall'
                      % (sample, n_total, cores, dt)}, list(st)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--n', type=int, default=65536, help='items per GPU')
    ap.add_argument('--cpu-sample', type=int, default=4096)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node == --gpus'
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the product has no CPU path'
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run: exercise RCCL init / barrier / all-reduce
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    import __graft_entry__ as ge
    pkg = ge.import_pkg()
    api = pkg.api
    lib = api.init(local_rank)
    n = args.n
    dev = torch.device('cuda', local_rank)

    # ---- synthetic inputs, signed on the device, left resident in HBM
    sks, msgs = gen_inputs(n, rank * n)
    skb = b''.join(s.to_bytes(32, 'little') for s in sks)
    d_msgs = torch.frombuffer(bytearray(b''.join(msgs)), dtype=torch.uint8).to(dev)
    d_offs = (torch.arange(n + 1, dtype=torch.int64) * 32).to(dev)
    d_pks = torch.empty(n * 288, dtype=torch.uint8, device=dev)
    d_sigs = torch.empty(n * 144, dtype=torch.uint8, device=dev)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    api._check(lib.blsgpu_sign_batch(1, api.POP, api._ptr(skb), P(d_msgs), P(d_offs), n, P(d_pks), P(d_sigs)))
    # negative controls: flip one bit of the message of 1 % of the items after signing
    bad = torch.arange(37, n, 100, device=dev)
    d_msgs[bad * 32] ^= 1
    expect = torch.zeros(n, dtype=torch.int32, device=dev)
    expect[bad] = api.INVALID_SIGNATURE
    torch.cuda.synchronize()

    def step():
        api._check(lib.blsgpu_verify_batch(1, api.POP, P(d_pks), P(d_sigs), P(d_msgs), P(d_offs), n, api.FMT_RAW_PROJ, P(d_status)))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    assert torch.equal(d_status, expect), 'verdict vector differs from the expected one'
    api.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = api.profile_read()
    api.profile_enable(False)
    assert torch.equal(d_status, expect), 'verdict vector differs from the expected one'
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        dom = max(prof.items(), key=lambda kv: kv[1][0])
        dom_ms = dom[1][0] / dom[1][1]
        achieved = ALG_BYTES_PER_VERIFY * n / (dom_ms * 1e-3) / 1e9
        kernels = {k: round(v[0] / v[1], 3) for k, v in prof.items()}
        out = {
            'metric': 'BLS12-381 sig verifications/sec (batch)', 'value': world * n * args.steps / dt, 'unit': 'verifications/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: %d independent Signature<Bls12381G1Impl>::verify items per GPU, 32-byte messages, '
                                   'PoP scheme, RAW_PROJ inputs resident in HBM, 1%% tampered' % n,
                       'items_per_gpu': n, 'sharding': 'independent batches per rank, no collective'},
            'roofline': {'bound': 'hbm', 'kernel': dom[0], 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'avg_launch_ms': dom_ms,
                         'algorithmic_bytes_per_launch': ALG_BYTES_PER_VERIFY * n,
                         'note': 'integer-VALU bound path: see valu_roofline; traffic is scratch (by-reference Fp12 operands), not input data'},
            'kernel_ms': kernels,
        }
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc):
            # HBM-side bytes per launch of the dominant kernel from rocprofv3 --pmc passes of this same command
            # (FETCH_SIZE and WRITE_SIZE in separate passes, KB units; FETCH_SIZE doubled: on gfx950 it tallies
            # 128-byte requests as 64 B -- MI355X_MICROARCH.md, HBM section).  Recorded by tools/pmc_traffic.py.
            t = json.load(open(pmc)).get(dom[0])
            if t:
                out['roofline']['traffic'] = (2 * t['FETCH_SIZE'] + t['WRITE_SIZE']) * 1024.0
                out['roofline']['traffic_source'] = t.get('source', 'profiles/pmc_traffic.json')
        fpm = os.path.join(ROOT, 'profiles', 'fpmul_counts.json')
        if os.path.exists(fpm):
            cnt = json.load(open(fpm))['verify_g1impl_fp_mul_equiv']
            rate = cnt * n * args.steps / dt / 1e9
            out['valu_roofline'] = {'fp_mul_per_verify': cnt, 'achieved': rate, 'peak': FPMUL_PEAK_G, 'unit': 'G fp_mul/s',
                                    'frac': rate / FPMUL_PEAK_G}
        if world == 1:
            sample = min(args.cpu_sample, n)
            msgs_tampered = list(msgs)
            for i in range(37, n, 100):
                msgs_tampered[i] = bytes([msgs[i][0] ^ 1]) + msgs[i][1:]
            out['cpu_baseline'], cpu_st = cpu_baseline(d_pks, d_sigs, msgs_tampered, sample, n)
            assert cpu_st == expect[:sample].cpu().tolist(), 'CPU oracle and GPU verdicts differ'  # checker, not measured path
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
