"""debug: step-by-step outputs of the sharded config-5 pipeline before / after a library verify_secure call"""
import ctypes, hashlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, bench
class A: gpus = 1
h = bench.Harness(A())
api, sh, ops = h.api, h.sh, h.ops
sg, n = 2, 65536
d_pks, d_sigs, _, _ = h.sign(sg, api.BASIC, n, 0, bench.FIXED_MSG * n, 32)
H = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]
def pipeline(tag):
    kb = ops.serialize(1, d_pks, n)
    perm = ops.sort_keys(kb, n, 48)
    dig = ops.keys_digest(kb, perm, n, 48)
    scal, st = ops.coefficients_for_range(dig, perm, n, 0, n)
    ident = torch.arange(n, dtype=torch.int32, device=h.dev)
    scal2, st2 = ops.coefficients_for_range(dig, ident, n, 0, n)
    first = ops.first_occurrence(kb, perm, n, 48)
    m1 = ops.point_sum(1, d_pks, n, scal)
    m2 = ops.point_sum(2, d_sigs, n, scal)
    s1 = ops.point_sum(1, d_pks, n)
    print(tag, 'kb', H(kb), 'perm', H(perm), 'dig', H(dig), 'scal', H(scal), st, 'scal2', H(scal2), 'first', H(first), 'msm1', H(ops.serialize(1, m1, 1)), 'msm2', H(ops.serialize(2, m2, 1)), 'sum', H(ops.serialize(1, s1, 1)), flush=True)
pipeline('run1')
pipeline('run2')
st, agg = sh.aggregate_secure(sg, d_pks, d_sigs, n, 0, 0, n_total=n)
stc = ctypes.c_int32(-9)
api._check(h.lib.blsgpu_verify_secure(sg, api.BASIC, h.P(d_pks), n, h.P(agg), api._ptr(bench.FIXED_MSG), 32, 0, 0, ctypes.byref(stc)))
print('lib verify_secure', stc.value)
pipeline('run3')
api.profile_enable(True)
api._check(h.lib.blsgpu_verify_secure(sg, api.BASIC, h.P(d_pks), n, h.P(agg), api._ptr(bench.FIXED_MSG), 32, 0, 0, ctypes.byref(stc)))
api.profile_read(); api.profile_enable(False)
print('lib verify_secure (profiled)', stc.value)
pipeline('run4')
