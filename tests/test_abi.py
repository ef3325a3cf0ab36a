"""The C-ABI library loads on a machine without a GPU, exports every symbol include/blsgpu.h declares, and every
compute entry point fails loudly there (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import util


@pytest.fixture(scope='module')
def lib(pkg):
    return pkg.api.load_library()


def declared_symbols():
    h = open(os.path.join(util.ROOT, 'include', 'blsgpu.h')).read()
    return sorted(set(re.findall(r'\b(blsgpu_\w+)\s*\(', h)))


def test_exports_every_declared_symbol(lib, pkg):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(pkg.api.EXPORTS) == names


def test_product_does_not_touch_the_oracle():
    """The product path must not import, include or link anything under oracle/."""
    root = os.path.join(util.ROOT, 'agora-blsful_amd')
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(('.py', '.cuh', '.hip', '.h', '.inc', '.cpp')):
                assert 'oracle' not in open(os.path.join(dp, f), errors='replace').read().replace('the oracle', ''), f


def test_fails_loudly_without_a_device(lib, pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    rc = lib.blsgpu_init(-1)
    assert rc == -1                                   # BLSGPU_E_NO_DEVICE
    buf = ctypes.create_string_buffer(256)
    lib.blsgpu_last_error(buf, 256)
    assert b'no CPU fallback' in buf.value
    st = ctypes.c_int32(0)
    assert lib.blsgpu_verify_batch(1, 2, None, None, None, None, 1, 0, ctypes.byref(st)) == -4      # NOT_INIT
    with pytest.raises(pkg.api.BlsGpuRuntimeError):
        pkg.api.verify_batch(1, 2, [bytes(288)], [bytes(144)], [b'm'])


def test_error_mapping(pkg):
    """status codes -> the reference's BlsError values and strings (src/traits/sig_core.rs:126-176, sig_basic.rs:51-55)."""
    f = pkg.api.error_from_status
    assert f(0) is None
    assert f(1) == pkg.BlsError('InvalidSignature')
    assert f(2) == pkg.BlsError('InvalidInputs', 'signature is the identity point')
    assert f(3) == pkg.BlsError('InvalidInputs', 'public key is the identity point')
    assert f(3, (5, 0), aggregate=True) == pkg.BlsError('InvalidInputs', 'public key at 5 is the identity point')
    assert f(4, (2, 9)) == pkg.BlsError('InvalidInputs', 'duplicate messages detected at 2 and 9')
    assert f(5) == pkg.BlsError('InvalidCoefficient')


def test_strict_env_refuses_unknown_and_ungated_variables():
    """BLSGPU_STRICT_ENV=1: blsgpu_init refuses an environment with a BLSGPU_ variable the knob table does not
    know, a non-integer value, or an A/B switch without BLSGPU_AB_KNOBS=1 -- before it looks for a device, so this runs on CPU.
    Without the strict mode the same environments reach the device probe (child processes: the knobs are parsed once)."""
    import subprocess
    import sys
    code = ("import ctypes, sys\n"
            "lib = ctypes.CDLL(%r)\n"
            "rc = lib.blsgpu_init(-1)\n"
            "buf = ctypes.create_string_buffer(512); lib.blsgpu_last_error(buf, 512)\n"
            "print(rc, buf.value.decode())\n") % os.path.join(util.ROOT, 'agora-blsful_amd', 'libblsgpu.so')
    import torch
    base = {k: v for k, v in os.environ.items() if not k.startswith('BLSGPU_')}

    def run(env):
        r = subprocess.run([sys.executable, '-c', code], env=dict(base, **env), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        rc, _, msg = r.stdout.strip().partition(' ')
        return int(rc), msg
    ok = (0,) if torch.cuda.is_available() else (-1,)           # BLSGPU_E_NO_DEVICE here, success on a GPU box
    assert run({'BLSGPU_STRICT_ENV': '1'})[0] in ok
    assert run({'BLSGPU_STRICT_ENV': '1', 'BLSGPU_CONTEXTS': '3', 'BLSGPU_LIB': 'x'})[0] in ok
    rc, msg = run({'BLSGPU_STRICT_ENV': '1', 'BLSGPU_CONTEXT': '3'})
    assert rc == -3 and 'unknown variable BLSGPU_CONTEXT' in msg
    rc, msg = run({'BLSGPU_STRICT_ENV': '1', 'BLSGPU_CONTEXTS': 'three'})
    assert rc == -3 and 'not an integer' in msg
    rc, msg = run({'BLSGPU_STRICT_ENV': '1', 'BLSGPU_MILLER_V1': '1'})
    assert rc == -3 and 'needs BLSGPU_AB_KNOBS=1' in msg
    assert run({'BLSGPU_STRICT_ENV': '1', 'BLSGPU_MILLER_V1': '1', 'BLSGPU_AB_KNOBS': '1'})[0] in ok
    assert run({'BLSGPU_CONTEXT': '3', 'BLSGPU_MILLER_V1': '1'})[0] in ok          # not strict: ignored (with a note on stderr), never fatal
