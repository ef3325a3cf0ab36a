"""Signature<Bls12381G2Impl>::verify as one 16,384-item batch: wall time of the call (host lists) and the kernel breakdown"""
import sys, time, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
pkg = ge.import_pkg(); api = pkg.api; api.init()
N = 16384
sks = [0x2222 + i for i in range(N)]
msgs = [hashlib.sha256(i.to_bytes(4, 'big')).digest() for i in range(N)]
for sg in (2, 1):
    pks, sigs = api.sign_batch(sg, api.POP, sks, msgs)
    assert not any(api.verify_batch(sg, api.POP, pks, sigs, msgs))
    api.profile_enable(True)
    t = time.perf_counter()
    for _ in range(3):
        st = api.verify_batch(sg, api.POP, pks, sigs, msgs)
    dt = (time.perf_counter() - t) / 3
    prof = api.profile_read(); api.profile_enable(False)
    print('sg', sg, 'n', N, 'call %.2f ms' % (dt * 1e3), {k: round(v[0] / v[1], 3) for k, v in prof.items()}, flush=True)
