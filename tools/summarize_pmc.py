#!/usr/bin/env python3
"""condense the counter_collection CSVs of rocprofv3 --pmc runs (one directory per pass under <dir>) into one small
table: mean per dispatch of one kernel (default mxe::chain_kernel_mc; second argument: another substring of the kernel name,
third: prefix of the pass directories, default pmc_), for every counter"""
import csv, glob, os, sys
root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else 'chain_kernel_mc'
prefix = sys.argv[3] if len(sys.argv) > 3 else 'pmc_'
rows = {}
for path in glob.glob(os.path.join(root, prefix + '*', '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel not in r.get('Kernel_Name', ''):
                continue
            name, val = r['Counter_Name'], float(r['Counter_Value'])
            rows.setdefault(name, []).append(val)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from maxent_amd import device
    print('#source_hash,%s' % device.source_hash())       # the build these counters belong to (bench.py checks it)
except Exception as exc:
    print('#source_hash,unknown (%r)' % (exc,))
print('counter,dispatches,mean_per_dispatch,min,max')
for name in sorted(rows):
    v = rows[name]
    # the first dispatches of a process run cold (L2, clocks): report all, the table says how many
    print('%s,%d,%.6g,%.6g,%.6g' % (name, len(v), sum(v) / len(v), min(v), max(v)))
