import sys, random
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import util
from util import c, ref
import __graft_entry__ as ge
pkg = ge.import_pkg(); api = pkg.api; api.init()
for C, sg in ((ref.G1Impl,1),(ref.G2Impl,2)):
    rng = random.Random(sg)
    pkraw, sigraw = (util.g2_raw, util.g1_raw) if sg == 1 else (util.g1_raw, util.g2_raw)
    sks=[ref.keygen_from_hash(bytes([11]) + i.to_bytes(4,'big') + bytes(27)) for i in range(5)]
    pks=[ref.public_key(C,s) for s in sks]
    msg=b'test message'
    sigs=[C.sig_curve.mul(C.hash_to_point(msg, C.DST[0]), s) for s in sks]
    agg=ref.aggregate_secure(C,pks,sigs,None)
    for use_rng in (False, True):
        R = rng if use_rng else None
        praw=[pkraw(p,R) for p in pks]
        kb=api.serialize(3-sg, praw)
        print(C.name, 'randZ', use_rng, 'compress ok', kb==[C.pk_to_bytes(p) for p in pks])
        st, perm, ts = api.secure_coefficients(kb)
        po,_,to = ref.secure_coefficients(kb)
        print('  coeff ok', (perm,ts)==(po,to))
        apk_dev = api.point_sum(3-sg, [praw[i] for i in perm], ts)
        apk=None
        for i,t in zip(po,to): apk=C.pk_curve.add(apk, C.pk_curve.mul(pks[i],t))
        print('  msm ok', api.serialize(3-sg,[apk_dev])[0]==C.pk_to_bytes(apk))
        print('  verify with dev apk', api.verify_batch(sg, 0, [apk_dev], [sigraw(agg,R)], [msg]))
        print('  verify_secure', api.verify_secure(sg, 0, praw, sigraw(agg,R), msg, 0))
