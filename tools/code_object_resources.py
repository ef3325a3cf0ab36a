#!/usr/bin/env python3
"""Per-kernel resources of the SHIPPED code objects (the private-segment figures quoted in DESIGN.md must come
from the library that travels, not from memory): unbundles every gfx950 code object of libblsgpu.so's .hip_fatbin section and
prints, per kernel, what its AMDGPU metadata note says -- VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, private segment (scratch
bytes per lane, with the non-inlined callees' share folded in by the compiler) and LDS bytes per workgroup -- plus instruction
counts of interest per code object (v_mad_i64_i32, DPP, ds_, scratch_, v_mfma).

usage: python tools/code_object_resources.py [path/to/libblsgpu.so] > profiles/r04_code_object_resources.txt
No GPU needed (llvm-objcopy / llvm-readelf / llvm-objdump from /opt/rocm/lib/llvm/bin)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.check_call([LLVM + '/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
        data = open(fat, 'rb').read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from('<Q', data, pos + 24)
        p = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx950' in triple and size:
                yield data[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernels_of(co_path):
    """[(name, {field: value})] from the msgpack-as-YAML view llvm-readelf prints"""
    txt = subprocess.run([LLVM + '/llvm-readelf', '--notes', co_path], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r'\n\s+- \.agpr_count:', txt)[1:]:
        blk = '.agpr_count:' + blk
        f = dict(re.findall(r'\.([a-z_]+):\s+(\S+)', blk))
        if 'name' in f:
            out.append((f['name'], f))
    return out


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r'\(.*', '', x) for x in r]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'agora-blsful_amd', 'libblsgpu.so')
    rows, census = [], []
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(code_objects(lib)):
            path = os.path.join(td, 'co%d.o' % k)
            open(path, 'wb').write(co)
            ks = kernels_of(path)
            names = demangle([n for n, _ in ks])
            for (n, f), dn in zip(ks, names):
                rows.append((dn, int(f.get('vgpr_count', 0)), int(f.get('agpr_count', 0)), int(f.get('sgpr_count', 0)), int(f.get('vgpr_spill_count', 0)),
                             int(f.get('sgpr_spill_count', 0)), int(f.get('private_segment_fixed_size', 0)), int(f.get('group_segment_fixed_size', 0)),
                             int(f.get('max_flat_workgroup_size', 0))))
            dis = subprocess.run([LLVM + '/llvm-objdump', '-d', '--mcpu=gfx950', path], capture_output=True, text=True).stdout
            census.append((names[0] if names else '?', len(ks), dis.count('v_mad_i64_i32'), len(re.findall(r'\b(?:row_|quad_perm|wave_|_dpp)', dis)),
                           len(re.findall(r'^\s+ds_', dis, re.M)), len(re.findall(r'^\s+scratch_', dis, re.M)), dis.count('v_mfma')))
    print('# per-kernel resources of %s (%d bytes), gfx950 code objects; tools/code_object_resources.py' % (os.path.relpath(lib, ROOT), os.path.getsize(lib)))
    print('%-46s %5s %5s %5s %7s %7s %9s %7s %5s' % ('kernel', 'vgpr', 'agpr', 'sgpr', 'vspill', 'sspill', 'private_B', 'lds_B', 'wg'))
    for r in sorted(rows):
        print('%-46s %5d %5d %5d %7d %7d %9d %7d %5d' % r)
    print()
    print('# instruction census per code object (first kernel names the unit)')
    print('%-46s %7s %13s %6s %6s %8s %6s' % ('unit', 'kernels', 'v_mad_i64_i32', 'dpp', 'ds_', 'scratch_', 'v_mfma'))
    for c in sorted(census):
        print('%-46s %7d %13d %6d %6d %8d %6d' % c)


if __name__ == '__main__':
    main()
