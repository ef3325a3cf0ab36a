"""MSM timing (device-resident points, 255-bit scalars) under the BLSGPU_MSM2_C / _CH / _Q overrides (BLSGPU_MSM_V1=1: the
first-generation bucket method).  usage: python tools/dbg/msm.py [n]   -- prints kernel totals per call and a result hash"""
import ctypes, sys, time, hashlib, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import __graft_entry__ as ge
import bench
pkg = ge.import_pkg(); api = pkg.api
ops = api.TensorOps(torch.device('cuda', 0))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
R = bench.R_ORDER
scal = torch.frombuffer(bytearray(b''.join((int.from_bytes(hashlib.sha256(i.to_bytes(4, 'big')).digest(), 'big') % R).to_bytes(32, 'little') for i in range(N))), dtype=torch.uint8).cuda()
off = (torch.arange(N + 1, dtype=torch.int64) * 1).cuda(); msgs = torch.zeros(N, dtype=torch.uint8).cuda()
for g in (2, 1):
    sg = 3 - g
    pks = torch.empty(N * (288 if g == 2 else 144), dtype=torch.uint8, device='cuda'); sigs = torch.empty(N * (144 if g == 2 else 288), dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    api._check(ops.lib.blsgpu_sign_batch(sg, api.POP, api._ptr(bench.sk_bytes(N, 0)), ops._p(msgs), ops._p(off), N, ops._p(pks), ops._p(sigs)))
    ops.point_sum(g, pks, N, scal)
    api.profile_enable(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        out = ops.point_sum(g, pks, N, scal)
    dt = (time.perf_counter() - t) / 3
    prof = api.profile_read(); api.profile_enable(False)
    print('G%d n=%d v1=%s c=%s CH=%s Q=%s: %.2f ms wall' % (g, N, os.environ.get('BLSGPU_MSM_V1'), os.environ.get('BLSGPU_MSM2_C'), os.environ.get('BLSGPU_MSM2_CH'), os.environ.get('BLSGPU_MSM2_Q'), dt * 1e3),
          {k: round(v[0] / 3, 3) for k, v in prof.items()}, hashlib.sha256(ops.serialize(g, out, 1).cpu().numpy().tobytes()).hexdigest()[:12], flush=True)
