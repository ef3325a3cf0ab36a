"""Timing only (experiment builds whose results may be wrong on purpose): the headline batch through blsgpu_verify_batch, per-kernel
times from the library's HIP events, NO verdict check.  usage: python tools/dbg/r4_time.py [impl 1|2] [steps]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument('impl', nargs='?', type=int, default=1)
ap.add_argument('steps', nargs='?', type=int, default=10)
a = ap.parse_args()
args = argparse.Namespace(gpus=1, backend='nccl', pg_timeout=180)
h = bench.Harness(args)
torch, api, lib = h.torch, h.api, h.lib
n = 65536
_, msgs = bench.gen_inputs(n, 0)
d_pks, d_sigs, d_msgs, d_offs = h.sign(a.impl, api.POP, n, 0, b''.join(msgs), 32)
d_status = torch.full((n,), -7, dtype=torch.int32, device=h.dev)
torch.cuda.synchronize()
def step():
    api._check(lib.blsgpu_verify_batch(a.impl, api.POP, h.P(d_pks), h.P(d_sigs), h.P(d_msgs), h.P(d_offs), n, api.FMT_RAW_PROJ, h.P(d_status)))
dt, prof = h.timed(step, a.steps, 3)
print('ms_per_step %.3f' % (dt / a.steps * 1e3), bench.kernel_ms(prof, a.steps), 'ok items', int((d_status == 0).sum().item()))
