"""verify_batch from 48/96-byte wire encodings (decompression + subgroup checks on the GPU), 65,536 items."""
import sys, time, hashlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
N=65536
sks=[0x1111+i for i in range(N)]; msgs=[hashlib.sha256(i.to_bytes(4,'big')).digest() for i in range(N)]
for sg in (1, 2):
    pks,sigs=api.sign_batch(sg, api.POP, sks, msgs)
    pkg_, sgg_ = (2, 1) if sg == 1 else (1, 2)
    cpk=api.serialize(pkg_, pks); csg=api.serialize(sgg_, sigs)
    st=api.verify_batch(sg, api.POP, cpk, csg, msgs, fmt=api.FMT_COMPRESSED)
    assert not any(st)
    api.profile_enable(True)
    t=time.perf_counter(); st=api.verify_batch(sg, api.POP, cpk, csg, msgs, fmt=api.FMT_COMPRESSED); dt=time.perf_counter()-t
    print('sig_group', sg, '%.1f ms wall' % (dt*1e3), {k: round(v[0]/v[1],3) for k,v in api.profile_read().items()})
    api.profile_enable(False)
