"""Host-side cost of the verify_secure coefficient derivation (sort + SHA-256 stream + n coefficient hashes), n = 65,536."""
import sys, time, os, hashlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; api.init()
n=65536
for width in (96, 48):
    keys=[hashlib.sha512(i.to_bytes(4,'big')).digest()[:48] * (width // 48) for i in range(n)]
    t=time.perf_counter(); st, perm, scal = api.secure_coefficients(keys); dt=time.perf_counter()-t
    print('width', width, 'status', st, '%.1f ms (includes Python marshalling)' % (dt*1e3))
