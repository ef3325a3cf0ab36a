"""The C++ host-side mirror of the reference types (include/blsful_hip.hpp) and its reference-shaped tests
(tests/cpp/mirror_test.cpp): built with g++ against the C ABI only (no torch, no Python in the process).

CPU: the mirror compiles, links against libblsgpu.so and FAILS LOUDLY without a device (no CPU path).
GPU: the whole C++ test program passes (reference tests/cpp_integration_test.rs:87-192,
tests/secure_aggregation_test.rs:143-235 and the scheme round trips)."""
import json
import os
import subprocess

import pytest

import util

ROOT = util.ROOT
LIBDIR = os.path.join(ROOT, 'agora-blsful_amd')


def build(tmp_path):
    exe = str(tmp_path / 'mirror_test')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'cpp', 'mirror_test.cpp'), '-L', LIBDIR, '-lblsgpu', '-Wl,-rpath,' + LIBDIR, '-o', exe])
    k = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ref_kats.json')))
    lines = ['cpp_message ' + k['cpp']['message'], 'cpp_naive_agg ' + k['cpp']['naive_agg_sig_pk12']]
    for name in ('sk', 'pk', 'sig'):
        lines += ['cpp_%s %s' % (name, h) for h in k['cpp'][name]]
    lines += ['prod57_sig ' + k['prod57']['sig'], 'prod57_message ' + k['prod57']['message']]
    lines += ['prod57_pk ' + h for h in k['prod57']['pks']]
    kat = str(tmp_path / 'kats.txt')
    open(kat, 'w').write('\n'.join(lines) + '\n')
    return exe, kat


def test_cpp_mirror_builds_and_needs_a_device(pkg, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by test_cpp_mirror_on_gpu')
    exe, kat = build(tmp_path)
    p = subprocess.run([exe, kat], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and 'no CPU fallback' in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(pkg, tmp_path):
    exe, kat = build(tmp_path)
    p = subprocess.run([exe, kat], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and ' 0 failed' in p.stdout, (p.returncode, p.stdout[-3000:], p.stderr[-2000:])
