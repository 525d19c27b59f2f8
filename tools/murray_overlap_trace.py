#!/usr/bin/env python3
"""Did the screen of chunk c + 1 really run beside the sweep of chunk c (MCX_OPT_MURRAY_OVERLAP)?  From a rocprofv3 kernel
trace of `tools/murray_overlap_ab.py <chunks> --trace`: of the time the chunked sweeps (k_remote_sweep_srow) were running,
how much had a k_screen_gemm running at the same moment, and how much wall time the pairs of kernels took together.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace -f csv -d /tmp/ovl -- python3 $REPO/tools/murray_overlap_ab.py 4 --trace
                  python3 $REPO/tools/murray_overlap_trace.py /tmp/ovl"""
import csv
import glob
import os
import sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = [r for f in files for r in csv.DictReader(open(f))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
gemm = [(a, b) for a, b, k in ev if "k_screen_gemm" in k]
sweep = [(a, b) for a, b, k in ev if "k_remote_sweep_srow" in k]
# only the big launches (the chunked passes): sweeps of at least 20 us
big = [(a, b) for a, b in sweep if b - a > 20000]
tot = sum(b - a for a, b in big)
ovl = 0
gi = 0
for a, b in big:
    for ga, gb in gemm:
        if gb <= a:
            continue
        if ga >= b:
            break
        ovl += max(0, min(b, gb) - max(a, ga))
print("kernels traced: %d screens, %d sweeps (%d of them > 20 us)" % (len(gemm), len(sweep), len(big)))
print("time in big sweeps %.2f ms, of it with a screen kernel running at the same moment: %.2f ms = %.1f %%"
      % (tot / 1e6, ovl / 1e6, 100.0 * ovl / max(tot, 1)))
