#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc / --kernel-trace CSV directory: per kernel name, mean counter values / durations."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'spmm'
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f'   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g} sum={sum(v):.4g}')
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Kernel_Name'][:90]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    for k, v in acc.items():
        print(f'{k}  n={len(v)} mean={sum(v)/len(v):.4f} ms  total={sum(v):.3f} ms')
