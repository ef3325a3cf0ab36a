"""latency of one row-wide Fp multiplication in a dependent chain on a lone wave (device time of blsgpu_debug_wide_mul / reps)"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
import util
pkg = ge.import_pkg(); api = pkg.api; api.init()
N = int(os.environ.get('WIDE_N', '4'))
a = [util.fp_raw(12345 + i) for i in range(N)]; b = [util.fp_raw(98765 + i) for i in range(N)]
api.debug_wide_mul(a, b, 10)
for reps in (1000, 20000):
    api.profile_enable(True)
    api.debug_wide_mul(a, b, reps)
    ms = api.profile_read()['k_wide'][0]
    api.profile_enable(False)
    print('reps %d: %.3f ms -> %.1f ns per multiplication' % (reps, ms, ms * 1e6 / reps))
