#!/usr/bin/env python3
"""Durations of the kernels matching the given patterns in a rocprofv3 kernel trace, grouped by grid.
usage: python tools/trace_by_grid.py <dir with *_kernel_trace.csv> pattern [pattern ...]"""
import collections
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)  # (the newest)
rows = list(csv.DictReader(open(f)))
for pat in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for r in rows:
        if pat in r['Kernel_Name']:
            key = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))
            agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(pat, "total %.1f ms in %d calls" % (tot / 1e3, sum(len(v) for v in agg.values())))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print("   grid", k, "calls", len(v), "avg %.1f us" % (sum(v) / len(v)), "min %.1f max %.1f" % (min(v), max(v)), "sum %.1f ms" % (sum(v) / 1e3))
