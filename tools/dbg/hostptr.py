"""PCIe-inclusive rate: blsgpu_verify_batch on 65,536 items handed over as HOST buffers (pageable memory, staged by the
library) against the same call on device-resident buffers."""
import ctypes, hashlib, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import __graft_entry__ as ge
pkg = ge.import_pkg(); api = pkg.api; lib = api.init(0)
n = 65536
sks = [0x5151 + i for i in range(n)]
msgs = [hashlib.sha256(i.to_bytes(4, 'big')).digest() for i in range(n)]
pks, sigs = api.sign_batch(1, api.POP, sks, msgs)
pkb, sgb, blob = b''.join(pks), b''.join(sigs), b''.join(msgs)
offs = (ctypes.c_uint64 * (n + 1))(*[32 * i for i in range(n + 1)])
st = (ctypes.c_int32 * n)()
V = lambda x: ctypes.cast(x, ctypes.c_void_p)
def host():
    api._check(lib.blsgpu_verify_batch(1, api.POP, api._ptr(pkb), api._ptr(sgb), api._ptr(blob), V(offs), n, 0, V(st)))
dev = torch.device('cuda', 0)
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (pkb, sgb, blob, bytes(offs))]
dst = torch.zeros(n, dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def device():
    api._check(lib.blsgpu_verify_batch(1, api.POP, P(d[0]), P(d[1]), P(d[2]), P(d[3]), n, 0, P(dst)))
for name, fn in (('host buffers', host), ('device buffers', device)):
    fn(); ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print('%s: %.2f ms per call, %.3f M verifications/s' % (name, 1e3 * min(ts), n / min(ts) / 1e6))
assert list(st) == [0] * n and not dst.any().item()
