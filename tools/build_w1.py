"""Builds agora-blsful_amd/libblsgpu_w1.so: the library with the lane-split kernels compiled for ONE wave per SIMD
(-DBLS_SPLIT_WAVES=1: 512-VGPR budget), the variant of the scratch-traffic experiment recorded in DESIGN.md section 5
(profiles/r02_pmc_w1.json).  Select it at run time with BLSGPU_LIB=<path>.  usage: python tools/build_w1.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'agora-blsful_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
sys.path.insert(0, ROOT)
from __graft_entry__ import TU_FLAGS  # noqa: E402  (the per-unit flags of the default build)


def main():
    out = os.path.join(ROOT, 'agora-blsful_amd', 'libblsgpu_w1.so')
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    with tempfile.TemporaryDirectory() as tmp:
        procs, objs = [], []
        for f in srcs:
            o = os.path.join(tmp, f[:-4] + '.o')
            objs.append(o)
            tpb = '512' if f == 'tu_wide.hip' else '64'
            procs.append(subprocess.Popen([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DBLS_SPLIT_WAVES=1',
                                           '--gpu-max-threads-per-block=' + tpb] + TU_FLAGS.get(f, []) + ['-c', os.path.join(CSRC, f), '-o', o]))
        for f, p in zip(srcs, procs):
            if p.wait() != 0:
                sys.exit('hipcc failed on ' + f)
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    print(out)


if __name__ == '__main__':
    main()
