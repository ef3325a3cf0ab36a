"""where the time of k_hash_to_g1_wide goes: the kernel launched through blsgpu_hash_to_g1 for one message, cut after each phase
(BLSGPU_HASH_STOP, read once per process: this script re-runs itself per phase)"""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    import __graft_entry__ as ge
    pkg = ge.import_pkg(); api = pkg.api; api.init()
    for _ in range(3):
        api.hash_to_point(1, [b'm' * 32], b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_')
    api.profile_enable(True)
    for _ in range(20):
        api.hash_to_point(1, [b'm' * 32], b'BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_POP_')
    ms, cnt = api.profile_read()['k_hash_to_point']
    print('stop %s: %.1f us' % (os.environ.get('BLSGPU_HASH_STOP', '-'), ms / cnt * 1e3), flush=True)
else:
    names = {1: 'expand + hash_to_field', 2: '+ SSWU (square-root chain)', 3: '+ 11-isogeny', 4: '+ point addition', 5: '+ 64 doublings, 5 additions', 0: 'whole hash'}
    for k in (1, 2, 3, 4, 5, 0):
        env = dict(os.environ, BLSGPU_HASH_STOP=str(k))
        out = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print('%-32s %s' % (names[k], out))
