import sys, os, time, ctypes, random
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
pkg=ge.import_pkg(); api=pkg.api; lib=api.init()
n=65536
for w in (96,48):
    rng=random.Random(1)
    blob=bytes(rng.getrandbits(8) for _ in range(n*w))
    perm=(ctypes.c_uint32*n)(); scal=ctypes.create_string_buffer(32*n); st=ctypes.c_int32()
    for _ in range(3):
        t=time.perf_counter()
        lib.blsgpu_secure_coefficients(api._ptr(blob), n, w, ctypes.cast(perm,ctypes.c_void_p), ctypes.cast(scal,ctypes.c_void_p), ctypes.byref(st))
        print(w, 'secure_coefficients %.1f ms' % ((time.perf_counter()-t)*1e3), st.value)
