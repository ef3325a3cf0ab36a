"""single Signature::verify latency through the C ABI (host pointers) + kernel breakdown"""
import ctypes, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
pkg = ge.import_pkg(); api = pkg.api; lib = api.init()
for sg in (1, 2):
    pks, sigs = api.sign_batch(sg, api.POP, [1234567, 7654321], [b'm' * 32, b'n' * 32])
    assert api.verify_batch(sg, api.POP, pks, sigs, [b'm' * 32, b'n' * 32]) == [0, 0]
    assert api.verify_batch(sg, api.POP, pks, sigs, [b'm' * 32, b'x' * 32]) == [0, 1]
    assert api.verify_batch(sg, api.POP, pks[::-1], sigs, [b'm' * 32, b'n' * 32]) == [1, 1]
    offs = (ctypes.c_uint64 * 2)(0, 32); st = ctypes.c_int32(-9)
    api.profile_enable(True)
    ts = []
    for _ in range(50):
        t = time.perf_counter()
        api._check(lib.blsgpu_verify_batch(sg, api.POP, api._ptr(pks[0]), api._ptr(sigs[0]), api._ptr(b'm' * 32), ctypes.cast(offs, ctypes.c_void_p), 1, 0, ctypes.cast(ctypes.byref(st), ctypes.c_void_p)))
        ts.append(time.perf_counter() - t)
    prof = api.profile_read(); api.profile_enable(False)
    ts.sort()
    print('sg', sg, 'single verify p50 %.3f ms' % (ts[25] * 1e3), {k: round(v[0] / v[1], 3) for k, v in prof.items()}, flush=True)
