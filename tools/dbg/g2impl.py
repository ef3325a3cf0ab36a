"""verify_batch throughput for Bls12381G2Impl (sig in G2, pk in G1: the Dash orientation), 65,536 items."""
import sys, time, hashlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
N=65536
sks=[0x1111+i for i in range(N)]; msgs=[hashlib.sha256(i.to_bytes(4,'big')).digest() for i in range(N)]
pks,sigs=api.sign_batch(2, api.POP, sks, msgs)
msgs2=list(msgs); msgs2[5]=b'x'
st=api.verify_batch(2, api.POP, pks, sigs, msgs2)
assert st[5]==1 and not any(st[:5]) and not any(st[6:])
api.profile_enable(True)
t=time.perf_counter(); st=api.verify_batch(2, api.POP, pks, sigs, msgs); dt=time.perf_counter()-t
print('G2Impl', N, 'items: %.1f ms wall (host staging included)' % (dt*1e3), {k: round(v[0]/v[1],3) for k,v in api.profile_read().items()})
