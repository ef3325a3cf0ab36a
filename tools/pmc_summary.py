"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel averages per launch.
usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections, csv, glob, json, sys
res = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if not (k.startswith('k_') or k.startswith('void k_')):
                continue
            k = k.split('(')[0].replace('void ', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            for c, x in v.items():
                res[k][c] = sum(x) / len(x)
print(json.dumps(res, indent=1, sort_keys=True))
