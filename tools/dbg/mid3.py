"""verify_batch wall time at mid sizes: the wave-cooperative pairing (default up to BLSGPU_COOP_MAX = 6144) against the lane-split kernels
(BLSGPU_COOP_MAX=<small>): re-runs itself per setting (the knob is read once per process)"""
import os, subprocess, sys, time, hashlib
if len(sys.argv) > 1:
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    import __graft_entry__ as ge
    api = ge.import_pkg().api; api.init()
    for n in (1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384):
        sks = [0x3333 + i for i in range(n)]
        msgs = [hashlib.sha256(i.to_bytes(4, 'big')).digest() for i in range(n)]
        pks, sigs = api.sign_batch(1, api.POP, sks, msgs)
        assert not any(api.verify_batch(1, api.POP, pks, sigs, msgs))
        t = time.perf_counter()
        for _ in range(3):
            api.verify_batch(1, api.POP, pks, sigs, msgs)
        print(sys.argv[1], n, '%.2f ms' % ((time.perf_counter() - t) / 3 * 1e3), flush=True)
else:
    for cm in ('6144', '512'):
        subprocess.check_call([sys.executable, __file__, 'coop_max=' + cm], env=dict(os.environ, BLSGPU_COOP_MAX=cm))
