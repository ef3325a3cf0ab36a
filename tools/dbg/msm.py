"""MSM timing (65,536 G2 / G1 points, 255-bit scalars) under the BLSGPU_MSM_C / BLSGPU_MSM_CH overrides."""
import sys, time, hashlib, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
N=65536
R=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
sks=[0x1111+i for i in range(N)]; msgs=[b'm']*N
pk2,_=api.sign_batch(1, api.POP, sks, msgs)     # G2 public keys
pk1,_=api.sign_batch(2, api.POP, sks, msgs)     # G1 public keys
scal=[int.from_bytes(hashlib.sha256(i.to_bytes(4,'big')).digest(),'big') % R for i in range(N)]
for g, pts in ((2, pk2), (1, pk1)):
    api.point_sum(g, pts, scal)
    api.profile_enable(True)
    t=time.perf_counter(); out=api.point_sum(g, pts, scal); dt=time.perf_counter()-t
    print('G%d c=%s CH=%s: %.1f ms wall' % (g, os.environ.get('BLSGPU_MSM_C'), os.environ.get('BLSGPU_MSM_CH'), dt*1e3), {k: round(v[0]/v[1],3) for k,v in api.profile_read().items()}, hashlib.sha256(out).hexdigest()[:12])
    api.profile_enable(False)
