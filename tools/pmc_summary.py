#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from a *_counter_collection.csv.
    python tools/pmc_summary.py <dir-or-csv> [substring]"""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
files = [path] if path.endswith('.csv') else glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if sub in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f'    {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})')
