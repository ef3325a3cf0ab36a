"""hash-to-G2 of one message (k_hash_to_g2, a lane pair): whole, and without the cofactor clearing (BLSGPU_HASH_STOP=9)"""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    import __graft_entry__ as ge
    pkg = ge.import_pkg(); api = pkg.api; api.init()
    dst = b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_'
    for _ in range(3):
        api.hash_to_point(2, [b'm' * 32], dst)
    api.profile_enable(True)
    for _ in range(10):
        api.hash_to_point(2, [b'm' * 32], dst)
    ms, cnt = api.profile_read()['k_hash_to_point']
    print('stop %s: %.1f us' % (os.environ.get('BLSGPU_HASH_STOP', '-'), ms / cnt * 1e3), flush=True)
else:
    for k in (9, 8, 0):
        env = dict(os.environ, BLSGPU_HASH_STOP=str(k))
        print(subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
