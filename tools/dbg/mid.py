"""verify_batch at mid-size batches under different BLSGPU_COOP_MAX thresholds."""
import sys, time, hashlib, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
N=16384
sks=[0x1111+i for i in range(N)]; msgs=[hashlib.sha256(i.to_bytes(4,'big')).digest() for i in range(N)]
pks,sigs=api.sign_batch(1, api.POP, sks, msgs)
for n in (2048, 4096, 6144, 8192, 12288, 16384):
    api.verify_batch(1, api.POP, pks[:n], sigs[:n], msgs[:n])
    api.profile_enable(True)
    st=api.verify_batch(1, api.POP, pks[:n], sigs[:n], msgs[:n])
    assert not any(st)
    pr=api.profile_read(); api.profile_enable(False)
    print(os.environ.get('BLSGPU_COOP_MAX'), n, round(sum(v[0] for v in pr.values()),2), {k: round(v[0]/v[1],2) for k,v in pr.items()}, flush=True)
