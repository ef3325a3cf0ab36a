"""Count Fp multiplications per verification (one REDC pass over one product = 1, the fused two-product pass = 1.5)
by instrumenting the host-compiled device headers (tests/hostsim) on the path the kernels run: non-split prepare, then
the lane-split Miller loop and final exponentiation.
Writes profiles/fpmul_counts.json, which bench.py uses for the integer-VALU roofline fraction."""
import ctypes, json, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import util
from util import ref
so = '/tmp/libhostsim_count.so'
subprocess.check_call(['g++', '-O1', '-DBLS_COUNT_FPMUL', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tests', 'hostsim', 'hostsim.cpp')])
hs = ctypes.CDLL(so)
cnt = ctypes.c_uint64.in_dll(hs, 'g_fpmul_halves')
ctypes.c_int.in_dll(hs, 'hs_device_path_only').value = 1
rng = random.Random(1)
res = {}
for name, C, sg, pkraw, sigraw in (('g1impl', ref.G1Impl, 1, util.g2_raw, util.g1_raw), ('g2impl', ref.G2Impl, 2, util.g1_raw, util.g2_raw)):
    sk = ref.keygen_from_hash(b'\x07' * 32); pk = ref.public_key(C, sk); m = bytes(32)
    sig = ref.sign(C, ref.POP, sk, m); dst = C.DST[ref.POP]
    cnt.value = 0
    st = hs.hs_verify(sg, pkraw(pk, rng), sigraw(sig, rng), 0, m, len(m), dst, len(dst))
    assert st == 0
    res['verify_%s_fp_mul_equiv' % name] = cnt.value // 2
    marks = (ctypes.c_uint64 * 3).in_dll(hs, 'hs_phase_marks')
    res['phases_%s' % name] = {'prepare': marks[0] // 2, 'miller': (marks[1] - marks[0]) // 2, 'finalexp': (marks[2] - marks[1]) // 2}
print(res)
json.dump(res, open(os.path.join(ROOT, 'profiles', 'fpmul_counts.json'), 'w'), indent=1)
