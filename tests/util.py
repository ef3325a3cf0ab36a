"""Test helpers: oracle <-> raw-buffer conversions (blst-layout Montgomery limbs) and library loading."""
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.py import bls381 as c  # noqa: E402
from oracle.py import blsful_ref as ref  # noqa: E402

P = c.P
RM = 1 << 384


def fp_raw(v):
    """Fp value -> 48 bytes, Montgomery form, little-endian limbs (blst `blst_fp`)."""
    return (v % P * RM % P).to_bytes(48, 'little')


def fp2_raw(v):
    return fp_raw(v[0]) + fp_raw(v[1])


def fp_from_raw(b):
    return int.from_bytes(b, 'little') * pow(RM, -1, P) % P


def g1_raw(pt, rng=None):
    """Affine oracle point -> 144-byte Jacobian RAW_PROJ (random Z when rng is given; Z = 0 for infinity)."""
    if pt is None:
        return fp_raw(0) + fp_raw(1) + fp_raw(0) if rng is None else fp_raw(rng.randrange(1, P)) + fp_raw(rng.randrange(1, P)) + fp_raw(0)
    z = 1 if rng is None else rng.randrange(1, P)
    return fp_raw(pt[0] * z * z) + fp_raw(pt[1] * z * z * z) + fp_raw(z)


def g2_raw(pt, rng=None):
    if pt is None:
        return fp2_raw((0, 0)) + fp2_raw((1, 0)) + fp2_raw((0, 0))
    z = (1, 0) if rng is None else (rng.randrange(1, P), rng.randrange(P))
    z2 = c.f2_sqr(z)
    z3 = c.f2_mul(z2, z)
    return fp2_raw(c.f2_mul(pt[0], z2)) + fp2_raw(c.f2_mul(pt[1], z3)) + fp2_raw(z)


def g1_aff_raw(pt):
    return fp_raw(pt[0]) + fp_raw(pt[1])


def g2_aff_raw(pt):
    return fp2_raw(pt[0]) + fp2_raw(pt[1])


def f12_from_plain_words(buf):
    """144 little-endian u32 words (6 Fp2 in w-power order, plain integers) -> oracle f12 tuple."""
    out = []
    for k in range(6):
        c0 = int.from_bytes(buf[96 * k:96 * k + 48], 'little')
        c1 = int.from_bytes(buf[96 * k + 48:96 * k + 96], 'little')
        out.append((c0, c1))
    return tuple(out)


def f12_raw(f):
    return b''.join(fp2_raw(x) for x in f)


def scalar_raw(k):
    return (k % (1 << 256)).to_bytes(32, 'little')


def build_hostsim():
    src = os.path.join(ROOT, 'tests', 'hostsim', 'hostsim.cpp')
    so = os.path.join(ROOT, 'tests', 'hostsim', 'libhostsim.so')
    if os.environ.get('BLS_HOSTSIM_SO'):          # a debug build (-O0 -g) for locating a tracked-bound violation
        return ctypes.CDLL(os.environ['BLS_HOSTSIM_SO'])
    deps = [src] + [os.path.join(ROOT, 'agora-blsful_amd', 'csrc', f) for f in os.listdir(os.path.join(ROOT, 'agora-blsful_amd', 'csrc')) if f.endswith('.cuh')]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        import subprocess
        # BLS_TRACK_BOUNDS: every Fp value carries worst-case limb/value bounds that are checked on every operation
        # (csrc/fp.cuh), so these tests also prove the placement of the carry/reduction passes for all inputs
        subprocess.check_call(['g++', '-O2', '-DBLS_TRACK_BOUNDS', '-shared', '-fPIC', '-o', so, src])
    return ctypes.CDLL(so)


def load_c_oracle():
    """Build (if needed) and load oracle/c/liboracle.so with argument types declared."""
    import subprocess
    d = os.path.join(ROOT, 'oracle', 'c')
    subprocess.check_call(['make', '-s', '-C', d])
    lib = ctypes.CDLL(os.path.join(d, 'liboracle.so'))
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    cp = ctypes.c_char_p
    lib.bo_init.restype = None
    lib.bo_verify.argtypes = [ci, ci, cp, cp, cp, sz]
    lib.bo_verify_batch.argtypes = [ci, ci, vp, vp, vp, vp, sz, vp, ci]
    lib.bo_verify_batch.restype = None
    lib.bo_hash_to_point.argtypes = [ci, cp, sz, cp, sz, vp]
    lib.bo_hash_to_point.restype = None
    lib.bo_compress.argtypes = [ci, cp, vp]
    lib.bo_compress.restype = None
    lib.bo_verify_secure.argtypes = [ci, ci, cp, sz, cp, cp, sz, ci]
    lib.bo_verify_secure_mt.argtypes = [ci, ci, vp, sz, vp, cp, sz, ci, ci]
    lib.bo_multi_verify.argtypes = [ci, ci, vp, sz, vp, cp, sz, ci]
    lib.bo_aggregate_verify.argtypes = [ci, ci, vp, vp, vp, sz, vp, ci, vp]
    lib.bo_init()
    return lib


_TOWER_TO_W = [0, 2, 4, 1, 3, 5]      # record slot (c0.a0, c0.a1, c0.a2, c1.a0, c1.a1, c1.a2) -> power of w


def f12_record(f):
    """oracle Fp12 (w-power order) -> the 576-byte record of the C ABI (tower order, Montgomery limbs)."""
    return b''.join(fp2_raw(f[_TOWER_TO_W[k]]) for k in range(6))


def f12_from_record(b):
    out = [None] * 6
    for k in range(6):
        out[_TOWER_TO_W[k]] = (fp_from_raw(b[96 * k:96 * k + 48]), fp_from_raw(b[96 * k + 48:96 * k + 96]))
    return tuple(out)
