"""MultiSignature::verify over 2^20 keys: time per call and kernel breakdown for the BLSGPU_ACC_LANES given in the environment"""
import sys, os, time, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import __graft_entry__ as ge
import bench
pkg = ge.import_pkg(); api = pkg.api; lib = api.init(0)
dev = torch.device('cuda', 0)
n = 1 << 20
P = lambda t: ctypes.c_void_p(t.data_ptr())
msg = bench.FIXED_MSG
d_msgs = torch.frombuffer(bytearray(msg * n), dtype=torch.uint8).to(dev)
d_offs = (torch.arange(n + 1, dtype=torch.int64) * 32).to(dev)
d_pks = torch.empty(n * 288, dtype=torch.uint8, device=dev); d_sigs = torch.empty(n * 144, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
api._check(lib.blsgpu_sign_batch(1, api.POP, api._ptr(bench.sk_bytes(n, 0)), P(d_msgs), P(d_offs), n, P(d_pks), P(d_sigs)))
agg = torch.empty(144, dtype=torch.uint8, device=dev)
api._check(lib.blsgpu_sum_g1(P(d_sigs), n, 0, P(agg)))
st = torch.zeros(1, dtype=torch.int32, device=dev)
def call():
    api._check(lib.blsgpu_multi_verify(1, api.POP, P(d_pks), n, P(agg), api._ptr(msg), 32, 0, ctypes.cast(ctypes.c_void_p(st.data_ptr()), ctypes.POINTER(ctypes.c_int32))))
for _ in range(3): call()
assert int(st.item()) == 0
ts = []
for _ in range(10):
    t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
ts.sort()
api.profile_enable(True)
for _ in range(3): call()
prof = api.profile_read(); api.profile_enable(False)
print('ACC_LANES', os.environ.get('BLSGPU_ACC_LANES'), 'p50 %.3f ms' % (ts[5] * 1e3), {k: round(v[0] / v[1], 3) for k, v in prof.items()}, flush=True)
