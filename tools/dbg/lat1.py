import sys, time, ctypes, hashlib, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
for n in (1, 64, 1024, 4096):
    sks=[0x1111+i for i in range(n)]; msgs=[hashlib.sha256(i.to_bytes(4,'big')).digest() for i in range(n)]
    pks,sigs=api.sign_batch(1, api.POP, sks, msgs)
    msgs2=list(msgs); msgs2[n//2]=b'x'
    st=api.verify_batch(1, api.POP, pks, sigs, msgs2)
    assert st==[0]*(n//2)+[1]+[0]*(n-n//2-1), st[:5]
    api.profile_enable(True)
    t=time.perf_counter(); api.verify_batch(1, api.POP, pks, sigs, msgs); dt=time.perf_counter()-t
    print(n, 'items: %.2f ms' % (dt*1e3), {k: round(v[0]/v[1],3) for k,v in api.profile_read().items()}, 'coop_max', os.environ.get('BLSGPU_COOP_MAX'))
    api.profile_enable(False)
