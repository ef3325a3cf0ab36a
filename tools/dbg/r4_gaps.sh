#!/bin/bash
# kernel start / end times of single verifications (tools/dbg/lat1.py) from a rocprofv3 kernel trace: the gaps between the kernels of
# one call's critical path.  Writes gpurun_out/r4b/gaps.txt
set -e -o pipefail
OUT=gpurun_out/r4b/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 tools/dbg/lat1.py > $OUT/lat1.txt 2> $OUT/lat1.err
python3 - $OUT > gpurun_out/r4b/gaps.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r.get('Queue_Id', '')))
for f in glob.glob(sys.argv[1] + '/**/*_memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', ''), ''))
rows.sort()
# the last 40 events of the first orientation's loop: find the last k_pairing_post2 of a run of G1Impl calls and print the events before it
idx = [i for i, r in enumerate(rows) if r[2] == 'k_pairing_post2']
for i in idx[20:23]:
    j = i
    while j > 0 and rows[j - 1][2] != 'k_pairing_post2':
        j -= 1
    t0 = rows[j][0]
    for s, e, n, q in rows[j:i + 3]:
        print('%9.1f %9.1f  %7.1f us  q%s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
    print()
PY
cat $OUT/lat1.txt
find $OUT -name "*_trace.csv" -delete
