"""debug: sharded-path (world 1) aggregate_secure / verify_secure vs the library's own entry points, sg = 2"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, bench
class A: gpus = 1
h = bench.Harness(A())
api, sh, ops = h.api, h.sh, h.ops
for sg in (1, 2):
    for n in (7, 1000, 1024, 5000, 65536):
        d_pks, d_sigs, _, _ = h.sign(sg, api.BASIC, n, 0, bench.FIXED_MSG * n, 32)
        st, agg = sh.aggregate_secure(sg, d_pks, d_sigs, n, 0, 0, n_total=n)
        out = torch.empty(144 if sg == 1 else 288, dtype=torch.uint8, device=h.dev)
        stc = ctypes.c_int32(-9)
        api._check(h.lib.blsgpu_aggregate_secure(sg, h.P(d_pks), h.P(d_sigs), n, 0, 0, h.P(out), ctypes.byref(stc)))
        same = bytes(ops.serialize(sg, agg, 1).cpu().numpy().tobytes()) == bytes(ops.serialize(sg, out, 1).cpu().numpy().tobytes())
        v1 = sh.verify_secure(sg, api.BASIC, d_pks, n, out, bench.FIXED_MSG, 0, 0, n_total=n)
        api._check(h.lib.blsgpu_verify_secure(sg, api.BASIC, h.P(d_pks), n, h.P(out), api._ptr(bench.FIXED_MSG), 32, 0, 0, ctypes.byref(stc)))
        print('sg', sg, 'n', n, 'aggregate_secure dist == lib:', same, 'verify dist:', v1, 'verify lib:', stc.value, flush=True)
