"""debug: which earlier config breaks config 5 / G2Impl in the same process"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
class A: gpus = 1
h = bench.Harness(A())
def t(name, fn):
    try:
        r = fn(); print(name, 'ok', round(r['ms_per_step'], 2), flush=True)
    except Exception as e:
        print(name, 'FAIL', type(e).__name__, e, flush=True)
seq = sys.argv[1]
for ch in seq:
    if ch == '3': t('c3', lambda: bench.run_config3(h, 1, 1, 65536))
    if ch == '4': t('c4', lambda: bench.run_config4(h, 1, 1, 8192))
    if ch == 'a': t('c5 g1m', lambda: bench.run_config5(h, 1, 1, 'g1m'))
    if ch == 'b': t('c5 g2m', lambda: bench.run_config5(h, 1, 1, 'g2m'))
    if ch == 'c': t('c5 g2l', lambda: bench.run_config5(h, 1, 1, 'g2l'))
