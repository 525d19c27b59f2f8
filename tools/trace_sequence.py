#!/usr/bin/env python3
"""The Murray kernels of one job in launch order, from a rocprofv3 kernel trace of a bench configuration (jobs are cut at
k_remote_draw: one per remote step).  usage: python tools/trace_sequence.py <trace dir> [remote steps per job] [job]"""
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
job = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_remote_prep' in r['Kernel_Name']]
tot = {}
for r in rows[idx[per * job]:idx[per * (job + 1)]]:
    n = r['Kernel_Name']
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    short = n.split('(')[0].replace('void ', '').replace('mcx::', '')
    tot[short] = tot.get(short, 0.0) + us
    if 'sweep' in n or 'gemm' in n:
        print("%-50s %8.1f us grid (%d,%d)" % (short[:50], us, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y'])))
    elif 'k_remote_prep' in n:
        print("--- remote step")
print("per job:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("   %8.1f us  %s" % (v, k))
