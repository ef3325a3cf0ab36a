#!/usr/bin/env python
"""Benchmark of the hot path: batch BLS12-381 signature verification on MI355X.

Default workload (BASELINE.json configs[1], the config the metric is quoted on): N independent (pk, msg, sig)
Signature<Bls12381G1Impl>::verify items per GPU (default 65,536, 32-byte messages, ProofOfPossession scheme = the
reference default), inputs already resident in HBM in RAW_PROJ form, 1 % of the items tampered as negative controls.
A "step" = one pass over the batch through the C ABI (blsgpu_verify_batch).  One process per GPU; N > 1 ranks shard
independent batches (weak scaling, no collective on the data path); timing = barrier + synchronize on both sides, max
over ranks.

--config 3 | 4 | 5 benches the other BASELINE configs instead, sharded over the ranks through agora-blsful_amd/dist.py
(device-resident shards, fixed-size RCCL all-gathers; STRONG scaling: the total size is BASELINE's, each rank holds 1/N):
  3  MultiSignature::verify, 1,048,576 G2 public keys, one message
  4  AggregateSignature::verify, 262,144 distinct (pk, msg) pairs, Basic scheme (duplicate-message rule on)
  5  verify_secure, 65,536 public keys (--variant g1m | g2m | g2l: Bls12381G1Impl Modern, Bls12381G2Impl Modern / Legacy)
The default run also times a few steps of configs 3-5 (and of config 2 for Bls12381G2Impl) AFTER the headline measurement and
reports them under "other_configs" of the same JSON line (--no-extras switches that off) AT EVERY N: `bench.py --gpus 8`
therefore measures BASELINE's 8-GPU configs (4 and 5) over RCCL too.  After each
of those entries the ranks agree on its outcome through the rendezvous store before anybody enters the next collective; a
failure ends the ride-along and every rank exits non-zero (--fail-extra rehearses that; --backend gloo rehearses the N > 1
control flow with several ranks on one card).

Launching: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` every process is one rank (RANK /
WORLD_SIZE in the environment).  A bare `python bench.py --gpus N` with N > 1 and no such environment starts exactly that launcher
as a CHILD process (before torch is imported, so the parent never touches a GPU), relays the child's one result line and exit code,
and kills the child's process group at --launch-timeout.  --gpus 1 never starts a child.

kernel_ms: {kernel: [device ms per STEP summed over its launches, launches per step]} from HIP events on the stream each kernel runs on
(blsgpu_profile_*); "tail_stream_overlapped" runs beside the other kernels and is not additive.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      the dominant kernel's algorithmic HBM bytes / its HIP-event-measured duration vs 8 TB/s (and vs 6.29 TB/s)
  cpu_baseline  the oracle timed on a bounded sample of the same workload on this box's host cores (config 2, N = 1)
"""
import argparse
import ctypes
import datetime
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
FIXED_MSG = hashlib.sha256(SEED + b'fixed').digest()
ALG_BYTES_PER_VERIFY = 468          # pk 288 + sig 144 + msg 32 + status 4
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0           # same guide: measured copy bandwidth (BASELINE.md section 4 asks for both)
FPMUL_PEAK_G = 74.0                 # profiles/r03_ubench3_mad_rates.txt: the fused two-product multiplier pass alone, chip-wide, 2-4 waves/SIMD
                                    # (72-76 G multiplication-equivalents/s over runs and boxes; round 1 measured 66.6 with lazy signed limbs)
FPMUL_MAD_ONLY_G = 85.0             # the same pipe with nothing but multiply-adds: 33.5 T lane-MAD/s / 392 per multiplication
CONFIG_SIZES = {3: 1048576, 4: 262144, 5: 65536}
VARIANTS = {'g1m': (1, 0, 'Bls12381G1Impl/Modern'), 'g2m': (2, 0, 'Bls12381G2Impl/Modern'), 'g2l': (2, 1, 'Bls12381G2Impl/Legacy')}


def gen_inputs(n, base):
    sks = [(S0 + base + i) % R_ORDER or 1 for i in range(n)]
    msgs = [hashlib.sha256(SEED + (base + i).to_bytes(8, 'little')).digest() for i in range(n)]
    return sks, msgs


def sk_bytes(n, base):
    """32-byte little-endian secret keys S0 + base + i (mod r) for i < n, without a Python loop over big ints."""
    import numpy as np
    out = np.zeros((n, 32), dtype=np.uint8)
    start = (S0 + base) % R_ORDER
    if start + n < R_ORDER and start > 0:   # no wrap (always, for these sizes): add the 64-bit counter to the low limbs
        lo = start & ((1 << 64) - 1)
        hi = start >> 64
        idx = np.arange(n, dtype=np.uint64)
        low = (np.uint64(lo) + idx)
        carry = (low < np.uint64(lo)).astype(np.uint64)
        out[:, :8] = low.view(np.uint8).reshape(n, 8)
        hi_rows = [(hi + c) for c in (0, 1)]
        hb = [np.frombuffer(h.to_bytes(24, 'little'), dtype=np.uint8) for h in hi_rows]
        out[:, 8:] = np.where(carry[:, None] == 0, hb[0][None, :], hb[1][None, :])
        return out.tobytes()
    return b''.join(((S0 + base + i) % R_ORDER or 1).to_bytes(32, 'little') for i in range(n))


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


class Harness:
    """Process group, library, device and the timing discipline shared by all configs."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        assert self.world == args.gpus, 'launch with torch.distributed.run --nproc-per-node == --gpus'
        assert torch.cuda.is_available(), 'bench.py needs a GPU: the product has no CPU path'
        # --backend gloo rehearses the N > 1 control flow on a box with fewer cards than ranks (ranks then share cards)
        ndev = torch.cuda.device_count()
        assert args.backend == 'gloo' or self.local_rank < ndev, 'one GPU per rank with --backend nccl'
        torch.cuda.set_device(self.local_rank % ndev)
        self.dev = torch.device('cuda', self.local_rank % ndev)
        self.backend = args.backend
        self.dist = None
        self.store = None
        if self.world > 1 or 'RANK' in os.environ:      # under torch.distributed.run: RCCL (or gloo) init / barrier / collectives
            import torch.distributed as dist_mod
            self.dist = dist_mod
            # a bounded timeout: a rank that fails outside a collective must not leave the others waiting for ever
            kw = {'device_id': self.dev} if args.backend == 'nccl' else {}
            self.dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=args.pg_timeout), **kw)
            get_store = getattr(dist_mod.distributed_c10d, '_get_default_store', None)    # private API: without it, agreement goes through an all_gather_object
            self.store = get_store() if get_store is not None else None
        self.pg_timeout = args.pg_timeout
        import __graft_entry__ as ge
        self.pkg = ge.import_pkg()
        self.api = self.pkg.api
        self.ops = self.api.TensorOps(self.dev)
        self.lib = self.ops.lib
        from agora_blsful_amd import dist as bd
        self.bd = bd
        self.sh = bd.Sharded(self.ops, self.dist)

    def P(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def rccl_ranks(self):
        """what the process group itself reports (1 without one): recorded in every result entry"""
        return self.dist.get_world_size() if self.dist is not None else 1

    def agree(self, tag, error):
        """Error agreement after a unit of work that contains collectives: every rank posts its outcome in the rendezvous STORE
        (TCP key-value, independent of the collective backend's state) and reads everybody's.  Returns the list of error
        strings (empty: all ranks succeeded).  A rank that never posts (dead, or stuck in a collective past the process-group
        timeout) makes the wait raise: the caller exits non-zero instead of hanging."""
        if self.store is None and self.dist is not None and self.world > 1:
            vals = [None] * self.world
            self.dist.all_gather_object(vals, error or 'ok')
            return ['rank %d: %s' % (r, v) for r, v in enumerate(vals) if v != 'ok']
        if self.store is None:
            return [error] if error else []
        self.store.set('agree/%s/%d' % (tag, self.rank), error or 'ok')
        keys = ['agree/%s/%d' % (tag, r) for r in range(self.world)]
        self.store.wait(keys, datetime.timedelta(seconds=self.pg_timeout + 30))
        vals = [self.store.get(k).decode() for k in keys]
        return ['rank %d: %s' % (r, v) for r, v in enumerate(vals) if v != 'ok']

    def timed(self, step, steps, warmup, after_warmup=None):
        """warmup untimed steps, then exactly `steps` steps between fences; (seconds = max over ranks, kernel profile)."""
        for _ in range(warmup):
            step()
        if after_warmup is not None and warmup > 0:
            after_warmup()
        self.api.profile_enable(True)
        self.sh.collective_s = 0.0
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        dt = time.perf_counter() - t0
        prof = self.api.profile_read()
        self.api.profile_enable(False)
        if self.dist is not None:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev if self.backend == 'nccl' else 'cpu')
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof

    def sign(self, sg, scheme, n, base, msgs_blob, msg_len):
        """device-resident (pks, sigs, msgs, offs) of the keys S0 + base + i, signed on the device"""
        torch = self.torch
        pksz, sgsz = (288, 144) if sg == 1 else (144, 288)
        d_msgs = torch.frombuffer(bytearray(msgs_blob), dtype=torch.uint8).to(self.dev) if n else torch.zeros(0, dtype=torch.uint8, device=self.dev)
        d_offs = (torch.arange(n + 1, dtype=torch.int64) * msg_len).to(self.dev)
        d_pks = torch.empty(max(n, 1) * pksz, dtype=torch.uint8, device=self.dev)
        d_sigs = torch.empty(max(n, 1) * sgsz, dtype=torch.uint8, device=self.dev)
        torch.cuda.synchronize()
        if n:
            self.api._check(self.lib.blsgpu_sign_batch(sg, scheme, self.api._ptr(sk_bytes(n, base)), self.P(d_msgs), self.P(d_offs), n, self.P(d_pks), self.P(d_sigs)))
        return d_pks[:n * pksz], d_sigs[:n * sgsz], d_msgs, d_offs

    def global_sum(self, group, local_partial):
        """fold of one group element per rank (setup only: builds the aggregate signatures the configs verify)"""
        parts = self.sh._all_gather(local_partial)
        return self.ops.point_sum(group, parts.reshape(-1), self.world)


OVERLAPPED = ('tail_stream_overlapped',)   # launched on a side stream beside the main stream's kernels: never "the dominant kernel"


def r6(x):
    """six significant digits (whole numbers stay exact): the driver keeps the last 8 KB of stdout, the line must fit"""
    return int(round(x)) if abs(x - round(x)) < 1e-9 * max(1.0, abs(x)) else float('%.6g' % x)


def kernel_ms(prof, steps):
    """{kernel: [device ms per step (all its launches), launches per step]}"""
    return {k: [round(v[0] / steps, 3), round(v[1] / steps, 2)] for k, v in prof.items() if v[1]}


def roofline_of(prof, steps, bytes_per_unit, units_per_step, kernel=None):
    """The dominant kernel = the largest per-STEP total among the kernels of the main stream.  One LAUNCH of it covers
    units_per_step / launches_per_step units (a chunked kernel covers one chunk), so achieved = bytes_per_unit x that / the
    average launch duration (the whole call's bytes over one chunk's launch was 4x too high for config 4)."""
    dom = max(((k, v) for k, v in prof.items() if k not in OVERLAPPED and v[1] and kernel in (None, k)), key=lambda kv: kv[1][0])
    launches = dom[1][1] / steps
    dom_ms = dom[1][0] / dom[1][1]
    units = units_per_step / launches
    alg = bytes_per_unit * units
    achieved = alg / (dom_ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': dom[0], 'achieved': r6(achieved), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': r6(achieved / HBM_PEAK_GBS),
            'frac_of_measured_copy_bw': r6(achieved / HBM_MEASURED_GBS), 'traffic': None, 'avg_launch_ms': r6(dom_ms),
            'launches_per_step': r6(launches), 'units_per_launch': r6(units), 'algorithmic_bytes_per_launch': r6(alg)}


# ------------------------------------------------------------------------------------------------- configs 3, 4, 5
def run_config3(h, steps, warmup, n_total=None, as_deserialised=False):
    """MultiSignature::verify (reference src/multi_signature.rs:127-135): n G2 keys, one message, sharded key sum.
    as_deserialised: the keys as a verifier that received them on the wire holds them (Z = 1: serialised and decompressed once,
    outside the timed region) instead of the device signer's Jacobian points with arbitrary Z."""
    api, sh = h.api, h.sh
    n = n_total or CONFIG_SIZES[3]
    lo, hi = h.bd.shard_range(n, h.rank, h.world)
    nl = hi - lo
    d_pks, d_sigs, _, _ = h.sign(1, api.POP, nl, lo, FIXED_MSG * nl, 32)
    if as_deserialised:
        wire = h.ops.serialize(2, d_pks, nl)
        st = h.torch.empty(nl, dtype=h.torch.int32, device=h.dev)
        h.torch.cuda.synchronize()
        api._check(h.lib.blsgpu_deserialize(2, h.P(wire), nl, api.FMT_COMPRESSED, h.P(d_pks), h.P(st)))
        assert int(st.abs().sum().item()) == 0
        del wire, st
    agg = h.global_sum(1, h.ops.point_sum(1, d_sigs, nl))
    del d_sigs
    bad = agg.clone()
    bad_msg = bytes([FIXED_MSG[0] ^ 1]) + FIXED_MSG[1:]
    assert sh.multi_verify(1, api.POP, d_pks, nl, agg, FIXED_MSG) == api.OK, 'config 3: the valid aggregate must verify'
    assert sh.multi_verify(1, api.POP, d_pks, nl, bad, bad_msg) == api.INVALID_SIGNATURE, 'config 3: negative control'
    res = []
    if h.world == 1:      # one process, one GPU: the library's own entry point (what the Rust shim calls)
        stc = ctypes.c_int32(-9)

        def step():
            api._check(h.lib.blsgpu_multi_verify(1, api.POP, h.P(d_pks), nl, h.P(agg), api._ptr(FIXED_MSG), 32, 0, ctypes.byref(stc)))
            res.append(stc.value)
    else:
        def step():
            res.append(sh.multi_verify(1, api.POP, d_pks, nl, agg, FIXED_MSG))
    dt, prof = h.timed(step, steps, warmup)
    assert all(r == api.OK for r in res)
    out = {'metric': 'MultiSignature::verify public keys/s', 'value': r6(n * steps / dt),
           'unit': 'public keys/s', 'ms_per_step': r6(dt / steps * 1e3), 'scaling': 'strong',
           'config': {'workload': 'configs[2] MultiSignature<Bls12381G1Impl>::verify, G2 keys RAW_PROJ%s in HBM, one 32-byte message, PoP'
                      % (' Z=1' if as_deserialised else ''), 'keys_total': n, 'keys_per_gpu': nl},
           'collective_ms_per_step': r6(sh.collective_s / steps * 1e3),
           'kernel_ms': kernel_ms(prof, steps)}
    if prof:
        out['roofline'] = roofline_of(prof, steps, 288, nl, 'k_accumulate' if prof.get('k_accumulate', (0, 0))[1] else None)   # the kernel that reads the keys
        out['whole_call_GBps'] = r6(288 * n * steps / dt / 1e9)
    return out


def run_config4(h, steps, warmup, n_total=None):
    """AggregateSignature::verify (reference src/aggregate_signature.rs:230-239), Basic scheme, distinct messages."""
    api, sh, torch = h.api, h.sh, h.torch
    n = n_total or CONFIG_SIZES[4]
    lo, hi = h.bd.shard_range(n, h.rank, h.world)
    nl = hi - lo
    blob = b''.join(hashlib.sha256(SEED + i.to_bytes(8, 'little')).digest() for i in range(lo, hi))
    d_pks, d_sigs, d_msgs, d_offs = h.sign(1, api.BASIC, nl, lo, blob, 32)
    agg = h.global_sum(1, h.ops.point_sum(1, d_sigs, nl))
    del d_sigs
    st, aux = sh.aggregate_verify(1, api.BASIC, d_pks, d_msgs, d_offs, nl, agg, lo, n_total=n)
    assert (st, tuple(aux)) == (api.OK, (0, 0)), ('config 4: the valid aggregate must verify', st, aux)
    # negative controls: one flipped message bit; a far-apart duplicate (reported with the reference's two indices)
    if h.rank == h.world - 1:
        d_msgs[(nl - 1) * 32] ^= 1
    st, _ = sh.aggregate_verify(1, api.BASIC, d_pks, d_msgs, d_offs, nl, agg, lo, n_total=n)
    assert st == api.INVALID_SIGNATURE, 'config 4: negative control'
    if h.rank == h.world - 1:
        d_msgs[(nl - 1) * 32] ^= 1
        keep = d_msgs[(nl - 1) * 32:nl * 32].clone()
        first = torch.frombuffer(bytearray(hashlib.sha256(SEED + (3).to_bytes(8, 'little')).digest()), dtype=torch.uint8).to(h.dev)
        d_msgs[(nl - 1) * 32:nl * 32] = first
    st, aux = sh.aggregate_verify(1, api.BASIC, d_pks, d_msgs, d_offs, nl, agg, lo, n_total=n)
    assert (st, tuple(aux)) == (api.DUPLICATE_MESSAGE, (3, n - 1)), ('config 4: duplicate rule', st, aux)
    if h.rank == h.world - 1:
        d_msgs[(nl - 1) * 32:nl * 32] = keep
    res = []
    if h.world == 1:      # one process, one GPU: the library's own entry point
        stc, aux = ctypes.c_int32(-9), (ctypes.c_uint64 * 2)()

        def step():
            api._check(h.lib.blsgpu_aggregate_verify(1, api.BASIC, h.P(d_pks), h.P(d_msgs), h.P(d_offs), nl, h.P(agg), 0, ctypes.byref(stc),
                                                     ctypes.cast(aux, ctypes.c_void_p)))
            res.append(stc.value)
    else:
        def step():
            res.append(sh.aggregate_verify(1, api.BASIC, d_pks, d_msgs, d_offs, nl, agg, lo, n_total=n)[0])
    dt, prof = h.timed(step, steps, warmup)
    assert all(r == api.OK for r in res)
    out = {'metric': 'AggregateSignature::verify (pk, msg) pairs/s', 'value': r6(n * steps / dt),
           'unit': 'pairs/s', 'ms_per_step': r6(dt / steps * 1e3), 'scaling': 'strong',
           'config': {'workload': 'configs[3] AggregateSignature<Bls12381G1Impl>::verify, distinct (pk, msg) pairs, Basic, 32-byte messages, RAW_PROJ in HBM',
                      'pairs_total': n, 'pairs_per_gpu': nl},
           'collective_ms_per_step': r6(sh.collective_s / steps * 1e3),
           'kernel_ms': kernel_ms(prof, steps)}
    if prof:
        out['roofline'] = roofline_of(prof, steps, 320, nl)
    return out


def run_config5(h, steps, warmup, variant='g1m', n_total=None):
    """verify_secure[_with_mode] (reference src/signature.rs:177-197,256-276, src/secure_aggregation.rs:173-208)."""
    api, sh = h.api, h.sh
    sg, mode, name = VARIANTS[variant]
    n = n_total or CONFIG_SIZES[5]
    lo, hi = h.bd.shard_range(n, h.rank, h.world)
    nl = hi - lo
    pk_group = 2 if sg == 1 else 1
    d_pks, d_sigs, _, _ = h.sign(sg, api.BASIC, nl, lo, FIXED_MSG * nl, 32)
    # the signature verify_secure accepts: sum t_i * sig_i (sign-side twin, checked against the oracle in tests/)
    st, agg = sh.aggregate_secure(sg, d_pks, d_sigs, nl, lo, mode, n_total=n)
    assert st == api.OK
    del d_sigs
    assert sh.verify_secure(sg, api.BASIC, d_pks, nl, agg, FIXED_MSG, lo, mode, n_total=n) == api.OK, 'config 5: must verify'
    bad_msg = bytes([FIXED_MSG[0] ^ 1]) + FIXED_MSG[1:]
    assert sh.verify_secure(sg, api.BASIC, d_pks, nl, agg, bad_msg, lo, mode, n_total=n) == api.INVALID_SIGNATURE, 'config 5: negative control'
    res = []
    if h.world == 1:      # one process, one GPU: the library's own entry point (what the Rust shim calls)
        stc = ctypes.c_int32(-9)

        def step():
            api._check(h.lib.blsgpu_verify_secure(sg, api.BASIC, h.P(d_pks), nl, h.P(agg), api._ptr(FIXED_MSG), 32, mode, 0, ctypes.byref(stc)))
            res.append(stc.value)
    else:
        def step():
            res.append(sh.verify_secure(sg, api.BASIC, d_pks, nl, agg, FIXED_MSG, lo, mode, n_total=n))
    dt, prof = h.timed(step, steps, warmup)
    assert all(r == api.OK for r in res)
    alg = (2 * (288 if pk_group == 2 else 144) + 64)
    out = {'metric': 'verify_secure public keys/s', 'value': r6(n * steps / dt), 'unit': 'public keys/s',
           'ms_per_step': r6(dt / steps * 1e3), 'scaling': 'strong', 'variant': name,
           'config': {'workload': 'configs[4] Signature<%s>::verify_secure%s, keys RAW_PROJ in HBM, Basic, one 32-byte message'
                                  % (name.split('/')[0], '_with_mode(Legacy)' if mode else ''), 'keys_total': n, 'keys_per_gpu': nl},
           'collective_ms_per_step': r6(sh.collective_s / steps * 1e3),
           'kernel_ms': kernel_ms(prof, steps)}
    if prof:
        out['roofline'] = roofline_of(prof, steps, alg, nl)
    return out


def run_config2_grouped(h, steps, warmup, n, tampered_every):
    """config 2 through the OPT-IN grouped entry (blsgpu_verify_batch_grouped: eight items share one final exponentiation through
    a seeded random linear combination; failing groups are re-verified item by item).  NOT the headline: the status vector equals
    blsgpu_verify_batch's only up to a 2^-64 chance per group that holds an invalid item.  tampered_every: 0 = all valid,
    k = every k-th item tampered (100 = the headline's 1 %); the per-item fallback of the failing groups is inside the time."""
    torch, api, lib, dev, P = h.torch, h.api, h.lib, h.dev, h.P
    _, msgs = gen_inputs(n, h.rank * n)
    d_pks, d_sigs, d_msgs, d_offs = h.sign(1, api.POP, n, h.rank * n, b''.join(msgs), 32)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
    expect = torch.zeros(n, dtype=torch.int32, device=dev)
    if tampered_every:
        bad = torch.arange(37 % tampered_every, n, tampered_every, device=dev)
        d_msgs[bad * 32] ^= 1
        expect[bad] = api.INVALID_SIGNATURE
    torch.cuda.synchronize()
    import secrets       # a fresh unpredictable seed per call, as include/blsgpu.h requires of every caller (advisor r3)

    def step():
        api._check(lib.blsgpu_verify_batch_grouped(1, api.POP, P(d_pks), P(d_sigs), P(d_msgs), P(d_offs), n, api.FMT_RAW_PROJ, secrets.randbits(64), P(d_status)))

    def check():
        assert torch.equal(d_status, expect), 'grouped verification: verdict vector differs from the expected one'

    dt, prof = h.timed(step, steps, warmup, after_warmup=check)
    check()
    return {'metric': 'BLS12-381 sig verifications/sec (batch, opt-in grouped mode)', 'value': r6(h.world * n * steps / dt), 'unit': 'verifications/s',
            'ms_per_step': r6(dt / steps * 1e3), 'scaling': 'weak',
            'config': {'workload': 'configs[1] through blsgpu_verify_batch_grouped (groups of 8 share one final exponentiation; %s; per-item fallback of failing groups included)'
                                   % ('every %d-th item tampered' % tampered_every if tampered_every else 'all valid'), 'items_per_gpu': n},
            'kernel_ms': kernel_ms(prof, steps)}


# ------------------------------------------------------------------------------------------------- config 2 (headline)
def run_config2(h, args, sg=1, steps=None, warmup=None, headline=True):
    """n independent Signature::verify items per GPU.  sg = 1: Bls12381G1Impl (BASELINE configs[1], the headline); sg = 2: the same
    batch for Bls12381G2Impl (keys in G1, signatures in G2: the orientation Dash and the reference's KATs use; reference
    src/impls/g2.rs:15-17,36-38) -- an other_configs entry.  headline: with roofline traffic, census fraction and the CPU leg."""
    torch, api, lib = h.torch, h.api, h.lib
    n, rank, world, dev = args.n, h.rank, h.world, h.dev
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    P = h.P
    # ---- synthetic inputs, signed on the device, left resident in HBM
    _, msgs = gen_inputs(n, rank * n)
    d_pks, d_sigs, d_msgs, d_offs = h.sign(sg, api.POP, n, rank * n, b''.join(msgs), 32)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
    # negative controls: flip one bit of the message of 1 % of the items after signing
    bad = torch.arange(37, n, 100, device=dev)
    d_msgs[bad * 32] ^= 1
    expect = torch.zeros(n, dtype=torch.int32, device=dev)
    expect[bad] = api.INVALID_SIGNATURE
    torch.cuda.synchronize()

    def step():
        api._check(lib.blsgpu_verify_batch(sg, api.POP, P(d_pks), P(d_sigs), P(d_msgs), P(d_offs), n, api.FMT_RAW_PROJ, P(d_status)))

    def check():
        assert torch.equal(d_status, expect), 'verdict vector differs from the expected one'

    dt, prof = h.timed(step, steps, warmup, after_warmup=check)
    check()
    out = None
    if rank == 0:
        impl = 'Bls12381G1Impl' if sg == 1 else 'Bls12381G2Impl'
        alg = ALG_BYTES_PER_VERIFY if sg == 1 else 144 + 288 + 32 + 4
        rl = roofline_of(prof, steps, alg, n)
        out = {
            'metric': 'BLS12-381 sig verifications/sec (batch)', 'value': r6(world * n * steps / dt), 'unit': 'verifications/s',
            'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': r6(dt / steps * 1e3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': 'configs[1] independent Signature<%s>::verify items, 32-byte messages, PoP, RAW_PROJ in HBM, 1%% tampered' % impl,
                       'items_per_gpu': n},
            'rccl_ranks': h.rccl_ranks(), 'collective_ms_per_step': 0.0,
            'roofline': rl,
            'kernel_ms': kernel_ms(prof, steps),
        }
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if headline and os.path.exists(pmc):
            # HBM-side bytes per launch of the dominant kernel from rocprofv3 --pmc passes of this same command
            # (FETCH_SIZE and WRITE_SIZE in separate passes, KB units; FETCH_SIZE doubled: on gfx950 it tallies
            # 128-byte requests as 64 B -- MI355X_MICROARCH.md, HBM section).  Recorded by tools/profile_round.sh.
            t = json.load(open(pmc)).get(rl['kernel'])
            if t:
                out['roofline']['traffic'] = r6((2 * t['FETCH_SIZE'] + t['WRITE_SIZE']) * 1024.0)
                out['roofline']['traffic_source'] = 'profiles/pmc_traffic.json'
            out['roofline']['note'] = 'integer-VALU bound path: see valu_roofline'

        fpm = os.path.join(ROOT, 'profiles', 'fpmul_counts.json')
        if os.path.exists(fpm):
            cnt = json.load(open(fpm))['verify_g1impl_fp_mul_equiv' if sg == 1 else 'verify_g2impl_fp_mul_equiv']
            rate = cnt * world * n * steps / dt / 1e9
            out['valu_roofline'] = {'fp_mul_per_verify': cnt, 'achieved_per_gpu': r6(rate / world), 'unit': 'G fp_mul/s',
                                    'peak': FPMUL_PEAK_G, 'frac': r6(rate / world / FPMUL_PEAK_G),
                                    'peak_mad_only': FPMUL_MAD_ONLY_G, 'frac_of_mad_only': r6(rate / world / FPMUL_MAD_ONLY_G)}
        if headline and world == 1:
            sample = min(args.cpu_sample, n)
            msgs_tampered = list(msgs)
            for i in range(37, n, 100):
                msgs_tampered[i] = bytes([msgs[i][0] ^ 1]) + msgs[i][1:]
            out['cpu_baseline'], cpu_st = cpu_baseline(d_pks, d_sigs, msgs_tampered, sample, n)
            assert cpu_st == expect[:sample].cpu().tolist(), 'CPU oracle and GPU verdicts differ'  # checker, not measured path
    return out


RESULT_FD = 1          # where the result line goes (__main__ moves stdout proper out of the libraries' reach)


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same
    args>` as a CHILD process group and relay its result line and exit code.  This process has not imported torch and never
    initialises a GPU; nothing is exec'ed.  The child group is killed at --launch-timeout."""
    import signal
    import socket
    import subprocess
    import threading
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = str(sock.getsockname()[1])
    sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + [('--items' if a == '--n' else a) for a in argv]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=2, env=env, cwd=ROOT, start_new_session=True)
    state = {'timed_out': False}

    def expire():
        state['timed_out'] = True
        sys.stderr.write('bench.py: the %d-rank child exceeded --launch-timeout %d s, killing its process group\n' % (args.gpus, args.launch_timeout))
        sys.stderr.flush()
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                return
            time.sleep(5)
    dog = threading.Timer(args.launch_timeout, expire)
    dog.daemon = True
    dog.start()
    lines = 0
    for raw in child.stdout:                       # the launcher passes the ranks' stdout through; rank 0 prints the one line
        if raw.lstrip().startswith(b'{'):
            os.write(RESULT_FD, raw if raw.endswith(b'\n') else raw + b'\n')
            lines += 1
        else:
            os.write(2, raw)
    rc = child.wait()
    dog.cancel()
    if state['timed_out'] and rc == 0:
        rc = 124
    if rc == 0 and lines == 0:
        sys.stderr.write('bench.py: the child exited 0 without a result line\n')
        rc = 4
    return rc if rc >= 0 else 128 - rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)      # with no flags: 25 steps of ~21 ms + the ride-along configs (at most 3 steps each): a few seconds
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--n', '--items', dest='n', type=int, default=65536,
                    help='items per GPU (config 2); under torch.distributed.run spell it --items: the launcher reads a bare --n as an abbreviation of its own options')
    ap.add_argument('--cpu-sample', type=int, default=4096)
    ap.add_argument('--config', type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument('--variant', default='g1m', choices=sorted(VARIANTS))
    ap.add_argument('--impl', default='g1', choices=['g1', 'g2'], help='config 2: Bls12381G1Impl (the headline, BASELINE configs[1]) or Bls12381G2Impl (profiling aid: implies --no-extras, no CPU leg)')
    ap.add_argument('--size', type=int, default=None, help='total size of config 3 / 4 / 5 (default: BASELINE.json\'s)')
    ap.add_argument('--keys-as-deserialised', action='store_true', help='config 3: feed keys with Z = 1 (what a verifier holds after decoding the wire bytes)')
    ap.add_argument('--grouped', type=int, default=None, metavar='K', help='config 2 through the opt-in grouped entry with every K-th item tampered (0: all valid); not the headline, not in the default run (frozen: DESIGN 9.5)')
    ap.add_argument('--no-extras', action='store_true', help='config 2 only: do not time configs 3-5 after the headline measurement')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend; gloo rehearses the N > 1 control flow with ranks sharing a card')
    ap.add_argument('--pg-timeout', type=int, default=180, help='process-group timeout in seconds')
    ap.add_argument('--extras-timeout', type=int, default=90, help='seconds ONE other_configs entry may take (input signing included; scaled up with the rank count) before the headline line is printed without the rest')
    ap.add_argument('--launch-timeout', type=int, default=1500, help='--gpus N > 1 without a launcher: seconds the child launcher may run before its process group is killed')
    ap.add_argument('--fail-extra', default=None, metavar='NAME[:RANK]', help='test aid: make that other_configs entry raise on that rank (default 0) before its collectives')
    args = ap.parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, argv))
    h = Harness(args)
    common = {'n_gpus': h.world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True, 'vs_baseline': None, 'dtype': 'u32',
              'data': 'synthetic'}
    exit_code = 0
    if args.config == 2 and args.grouped is not None:
        out = run_config2_grouped(h, args.steps, args.warmup, args.n, args.grouped)
        out.update(common)
    elif args.config == 2 and args.impl == 'g2':
        out = run_config2(h, args, sg=2, headline=False)
        if out is not None:
            out['config']['note'] = 'NOT the headline: Bls12381G2Impl orientation of configs[1] (--impl g2)'
    elif args.config == 2:
        out = run_config2(h, args)
        if not args.no_extras:
            # The other BASELINE configs ride along at every N: at N > 1 they are collectives over RCCL, so after each one the
            # ranks AGREE on its outcome through the rendezvous store (Harness.agree) before anybody enters the next collective.
            # An entry that failed on some rank is recorded as an error and ends the ride-along: the headline line is still
            # printed, every rank then exits non-zero (a collective that lost a rank ends at the process-group timeout).
            extras = {}
            k = max(1, min(args.steps, 3))
            size = lambda c: min(CONFIG_SIZES[c], args.size) if args.size else None  # noqa: E731
            plan = [('config3_multi_verify_1048576', lambda: run_config3(h, k, 1, size(3))),
                    ('config4_aggregate_verify_262144', lambda: run_config4(h, k, 1, size(4))),
                    ('config5_verify_secure_65536_g1impl_modern', lambda: run_config5(h, k, 1, 'g1m', size(5))),
                    ('config5_verify_secure_65536_g2impl_modern', lambda: run_config5(h, k, 1, 'g2m', size(5))),
                    ('config5_verify_secure_65536_g2impl_legacy', lambda: run_config5(h, k, 1, 'g2l', size(5))),
                    ('config2_g2impl_65536', lambda: run_config2(h, args, sg=2, steps=k, warmup=1, headline=False))]
            fail_name, _, fail_rank = (args.fail_extra or '').partition(':')
            # A watchdog PER ENTRY (re-armed; advisor r3: one timer over the whole ride-along would kill a healthy but slow N = 8
            # run): whatever happens in an entry (a collective that never returns, a backend that aborts the process at ITS
            # timeout), the headline line gets printed -- with the entries finished so far -- before this process leaves.  Every
            # rank runs the same timer, so all of them exit.  `written` (under a lock) makes the line go out exactly once.
            import threading
            state = {'current': None, 'written': False}
            wlock = threading.Lock()
            per_entry = args.extras_timeout * (1 + h.world // 4)

            def write_result(o):
                with wlock:
                    if state['written']:
                        return
                    state['written'] = True
                    if h.rank == 0 and o is not None:
                        os.write(RESULT_FD, (json.dumps(o, separators=(',', ':')) + '\n').encode())

            def expire(name, done):
                o = dict(out) if out is not None else None
                if o is not None:
                    done = dict(done)
                    done[name] = {'error': 'timeout: this other_configs entry exceeded %d s' % per_entry}
                    o['other_configs'] = done
                write_result(o)
                sys.stderr.write('bench.py rank %d: other_configs watchdog fired in %s, exiting 3\n' % (h.rank, name))
                sys.stderr.flush()
                os._exit(3)
            for name, fn in plan:
                state['current'] = name
                dog = threading.Timer(per_entry, expire, (name, dict(extras)))
                dog.daemon = True
                dog.start()
                err, r = None, None
                try:
                    if fail_name == name and h.rank == int((fail_rank or '0').split(':')[0]):
                        if fail_rank.endswith(':hang'):
                            time.sleep(10 ** 6)
                        raise RuntimeError('--fail-extra: deliberate failure before the collectives of this entry')
                    r = fn()
                except Exception as e:  # noqa: BLE001 -- the headline line must survive a failing extra
                    err = '%s: %s' % (type(e).__name__, e)
                errs = h.agree(name, err)
                dog.cancel()
                if errs:
                    extras[name] = {'error': '; '.join(errs)[:300]}
                    exit_code = 3
                    break
                if r is not None:
                    for drop in ('higher_is_better', 'vs_baseline', 'dtype', 'data', 'valu_roofline', 'warmup'):
                        r.pop(drop, None)
                    r.update({'n_gpus': h.world, 'steps': k, 'rccl_ranks': h.rccl_ranks()})
                    r.setdefault('collective_ms_per_step', 0.0)
                    extras[name] = r
                h.torch.cuda.empty_cache()
            if out is not None:
                out['other_configs'] = extras
            write_result(out)
            out = None                  # written (exactly once, also against a late watchdog)
    else:
        if args.config == 3:
            out = run_config3(h, args.steps, args.warmup, args.size, args.keys_as_deserialised)
        elif args.config == 4:
            out = run_config4(h, args.steps, args.warmup, args.size)
        else:
            out = run_config5(h, args.steps, args.warmup, args.variant, args.size)
        out.update(common)
        out['rccl_ranks'] = h.rccl_ranks()
    if h.rank == 0 and out is not None:
        os.write(RESULT_FD, (json.dumps(out, separators=(',', ':')) + '\n').encode())
    if exit_code:
        # ranks may be stuck in (or have timed out of) a collective: no orderly teardown, just leave -- non-zero, all of them
        sys.stderr.write('bench.py rank %d: an other_configs entry failed on some rank, exiting %d\n' % (h.rank, exit_code))
        sys.stderr.flush()
        os._exit(exit_code)
    if h.dist is not None:
        h.dist.destroy_process_group()


if __name__ == '__main__':
    # stdout carries ONE line, the result: whatever libraries write there (RCCL prints a five-line version banner on stdout when
    # its communicator comes up) goes to stderr instead
    sys.stdout.flush()
    RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    main()
