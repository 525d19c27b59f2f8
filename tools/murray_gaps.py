#!/usr/bin/env python3
"""Where a Murray job's stream sits idle: from a rocprofv3 kernel trace of `bench.py --config c3-murray|c5` (the directories
tools/refresh_profiles.sh leaves under gpurun_out/<round>_kt_*), per job (cut at k_run_reset, warm-ups skipped): span, busy
time, idle time, and the idle time by the kernel that ends each gap -- k_cull_stats / k_screen_prep_x / k_remote_draw_multi
start a pass after the HOST has read the survivors' count, k_square ends genRemote; gaps between kernels queued back to
back are dispatch latency (inflated under the profiler).  usage: murray_gaps.py gpurun_out/r05_kt_c3m [...]"""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
    starts = [i for i, e in enumerate(ev) if "k_run_reset" in e[2]]
    spans, busys, idles, counts = [], [], [], []
    byk = defaultdict(lambda: [0, 0])
    for a, b in zip(starts[5:-1], starts[6:]):
        job = ev[a:b]
        cur, busy, idle = job[0][0], 0, 0
        for s, e, k in job:
            if s > cur:
                idle += s - cur
                kk = k.split("(")[0].replace("void ", "").replace("mcx::", "")
                byk[kk][0] += s - cur
                byk[kk][1] += 1
            busy += max(0, e - max(s, cur))
            cur = max(cur, e)
        spans.append(job[-1][1] - job[0][0]); busys.append(busy); idles.append(idle); counts.append(len(job))
    n = len(spans)
    print("%s: %d jobs, span %.3f ms, busy %.3f ms, idle %.3f ms (%.1f %%), %d kernels per job"
          % (d, n, statistics.mean(spans) / 1e6, statistics.mean(busys) / 1e6, statistics.mean(idles) / 1e6,
             100.0 * statistics.mean(idles) / statistics.mean(spans), statistics.mean(counts)))
    for k, (t, c) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:10]:
        print("   idle before %-46s %.3f ms per job in %4.1f gaps (%.1f us each)" % (k, t / 1e6 / n, c / n, t / 1e3 / max(c, 1)))
