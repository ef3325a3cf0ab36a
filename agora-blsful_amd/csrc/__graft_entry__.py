"""Driver entry points: build() compiles every HIP extension for gfx950 (and the oracle's C restatement);
smoke() runs one small invocation of the hot path on cuda:0 and checks it against the oracle."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, 'agora-blsful_amd')
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB = os.path.join(PKG_DIR, 'libblsgpu.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def import_pkg():
    """Import the package directory `agora-blsful_amd/` under the module name `agora_blsful_amd`."""
    if 'agora_blsful_amd' in sys.modules:
        return sys.modules['agora_blsful_amd']
    spec = importlib.util.spec_from_file_location('agora_blsful_amd', os.path.join(PKG_DIR, '__init__.py'),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules['agora_blsful_amd'] = mod
    spec.loader.exec_module(mod)
    return mod


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    """Compile libblsgpu.so for gfx950 in-tree (hipcc cross-compiles without a GPU) and import the package."""
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, 'include', 'blsgpu.h')]
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        if force or _newer(o, deps):
            procs.append((s, subprocess.Popen([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', s, '-o', o])))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + s)
    if force or procs or _newer(LIB, objs):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    oracle_mk = os.path.join(ROOT, 'oracle', 'c', 'Makefile')
    if os.path.exists(oracle_mk):
        subprocess.check_call(['make', '-s', '-C', os.path.dirname(oracle_mk)])
    pkg = import_pkg()
    pkg.api.load_library()           # dlopen + symbol check; needs no device
    return pkg


def smoke():
    """One small verify_batch on cuda:0 (valid + tampered + identity cases) checked against the oracle."""
    import random
    import torch
    assert torch.cuda.is_available(), 'smoke() needs a GPU'
    torch.cuda.init()
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import util
    from oracle.py import blsful_ref as ref
    pkg = import_pkg()
    rng = random.Random(11)
    C = ref.G1Impl
    pks, sigs, msgs, expect = [], [], [], []
    for i in range(8):
        sk = ref.keygen_from_hash(bytes([i]) * 32)
        pk = ref.public_key(C, sk)
        m = b'smoke-%d' % i
        sig = ref.sign(C, ref.POP, sk, m)
        if i % 3 == 1:
            m = m + b'!'
        pks.append(util.g2_raw(pk, rng))
        sigs.append(util.g1_raw(sig if i != 5 else None, rng))
        msgs.append(m)
        try:
            ref.verify(C, ref.POP, pk, sig if i != 5 else None, m)
            expect.append(0)
        except ref.BlsError as e:
            expect.append({'InvalidSignature': 1}.get(e.kind, 2 if 'signature is' in e.msg else 3))
    got = pkg.api.verify_batch(1, pkg.api.POP, pks, sigs, msgs)
    assert got == expect, (got, expect)
    print('smoke ok: statuses', got)


if __name__ == '__main__':
    build()
    if len(sys.argv) > 1 and sys.argv[1] == 'smoke':
        smoke()
