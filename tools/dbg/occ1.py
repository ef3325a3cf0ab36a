"""Kernel time vs batch size around the one-round occupancy point (65536 items = 2048 waves = 2 per SIMD)."""
import sys, time, hashlib, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
N=int(sys.argv[1]) if len(sys.argv)>1 else 65536
print('start', N, flush=True)
sks=[0x1111+i for i in range(N)]; msgs=[hashlib.sha256(i.to_bytes(4,'big')).digest() for i in range(N)]
t0=time.perf_counter(); pks,sigs=api.sign_batch(1, api.POP, sks, msgs); print('signed %.1f s' % (time.perf_counter()-t0), flush=True)
for n in (8192, 16384, 32768, 49152, 61440, 65536, 69632, 81920, 98304, 131072):
    if n > N: break
    api.verify_batch(1, api.POP, pks[:n], sigs[:n], msgs[:n])
    api.profile_enable(True)
    t=time.perf_counter(); st=api.verify_batch(1, api.POP, pks[:n], sigs[:n], msgs[:n]); dt=time.perf_counter()-t
    assert not any(st)
    print(n, 'items: %.2f ms' % (dt*1e3), {k: round(v[0]/v[1],3) for k,v in api.profile_read().items()}, flush=True)
    api.profile_enable(False)
