"""Kernel time vs batch size near full occupancy (scratch footprint vs the 256 MB Infinity Cache)."""
import sys, time, hashlib, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
N=65536
sks=[0x1111+i for i in range(N)]; msgs=[hashlib.sha256(i.to_bytes(4,'big')).digest() for i in range(N)]
t0=time.perf_counter(); pks,sigs=api.sign_batch(1, api.POP, sks, msgs); print('signed %.1f s' % (time.perf_counter()-t0), flush=True)
for n in (int(x) for x in os.environ.get("OCC_NS", "32768,40960,49152,53248,57344,61440,65536").split(",")):
    api.verify_batch(1, api.POP, pks[:n], sigs[:n], msgs[:n])
    api.profile_enable(True)
    st=api.verify_batch(1, api.POP, pks[:n], sigs[:n], msgs[:n])
    assert not any(st)
    print(n, {k: round(v[0]/v[1],3) for k,v in api.profile_read().items()}, flush=True)
    api.profile_enable(False)
