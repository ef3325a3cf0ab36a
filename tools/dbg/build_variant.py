"""Build agora-blsful_amd/libblsgpu_<tag>.so = the current objects with the named translation units recompiled with extra flags
(A/B runs of per-unit compiler flags: BLSGPU_LIB=<that file> python bench.py ...).
usage: python tools/dbg/build_variant.py <tag> <unit>[,<unit>...] -- <flags...>     (units without the .hip suffix)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
tag, units = sys.argv[1], sys.argv[2].split(',')
flags = sys.argv[sys.argv.index('--') + 1:]
ge.build()
objs = []
for f in sorted(os.listdir(ge.CSRC)):
    if not f.endswith('.hip'):
        continue
    base = f[:-4]
    o = os.path.join(ge.CSRC, base + '.o')
    if base in units:
        o = os.path.join(ge.CSRC, base + '.' + tag + '.o')
        tpb = '512' if f == 'tu_wide.hip' else '64'
        subprocess.check_call([ge.HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--gpu-max-threads-per-block=' + tpb] + ge.COMMON_FLAGS +
                              ge.TU_FLAGS.get(f, []) + flags + ['-c', os.path.join(ge.CSRC, f), '-o', o])
    objs.append(o)
out = os.path.join(ge.PKG_DIR, 'libblsgpu_%s.so' % tag)
subprocess.check_call([ge.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
print(out)
