"""verify_batch wall time at small batch sizes under BLSGPU_WIDE_MAX (the largest batch that takes the row-wide engine); usage:
python tools/dbg/small.py [sig_group, default 1]"""
import sys, time, hashlib, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
pkg = ge.import_pkg(); api = pkg.api; api.init()
N = 2048
sks = [0x1111 + i for i in range(N)]; msgs = [hashlib.sha256(i.to_bytes(4, 'big')).digest() for i in range(N)]
SG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pks, sigs = api.sign_batch(SG, api.POP, sks, msgs)
SIZES = [int(x) for x in os.environ.get('SMALL_SIZES', '1,16,64,128,256,384,512,768,1024,2048').split(',')]
for n in SIZES:
    api.verify_batch(SG, api.POP, pks[:n], sigs[:n], msgs[:n])
    ts = []
    for _ in range(15 if n <= 64 else 5):
        t = time.perf_counter(); st = api.verify_batch(SG, api.POP, pks[:n], sigs[:n], msgs[:n]); ts.append(time.perf_counter() - t)
    assert not any(st)
    print(os.environ.get('BLSGPU_WIDE_MAX'), n, 'ms %.3f' % (min(ts) * 1e3), flush=True)
