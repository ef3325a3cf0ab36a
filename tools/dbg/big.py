"""verify_batch beyond one round of resident waves: 200,000 items, every 997th message tampered, both orientations."""
import sys, time, hashlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
N=200000
sks=[0x2222+i for i in range(N)]; msgs=[hashlib.sha256(i.to_bytes(4,'big')).digest() for i in range(N)]
for sg in (1, 2):
    pks,sigs=api.sign_batch(sg, api.POP, sks, msgs)
    m2=list(msgs); exp=[0]*N
    for i in range(3, N, 997):
        m2[i]=bytes([m2[i][0]^1])+m2[i][1:]; exp[i]=1
    t=time.perf_counter(); st=api.verify_batch(sg, api.POP, pks, sigs, m2); dt=time.perf_counter()-t
    assert st==exp, [i for i in range(N) if st[i]!=exp[i]][:5]
    print('sig_group', sg, N, 'items ok, %.1f ms wall (host lists + staging included)' % (dt*1e3))
