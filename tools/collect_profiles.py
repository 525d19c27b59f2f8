#!/usr/bin/env python3
"""Turn a round's rocprofv3 output under gpurun_out/ into the tracked summaries under profiles/.

  tools/collect_profiles.py r02

expects (all optional, per configuration C in c2 c3 c5 c3-murray):
  gpurun_out/<rnd>_bench_C.json   the JSON line of `bench.py --config C [--keep-pmc gpurun_out/<rnd>_pmc_C]`
  gpurun_out/<rnd>_pmc_C/<COUNTERS>/**/*counter_collection.csv   the live PMC passes bench.py kept
  gpurun_out/<rnd>_kt_C/**/*kernel_stats.csv    `rocprofv3 --kernel-trace --stats -- python3 bench.py --config C ...`
(gpurun merges into gpurun_out/ without deleting: remove gpurun_out/<rnd>_pmc_* and <rnd>_kt_* before a new
`gpurun -- bash tools/refresh_profiles.sh <rnd>`, or counters of kernels that no longer exist stay in the tables)
and writes
  profiles/<rnd>_bench_C.json, profiles/<rnd>_C_kernel_stats.csv,
  profiles/<rnd>_C_pmc_counters.json  (per kernel: mean per dispatch of every collected counter, second job only),
  profiles/<rnd>_C_fused_kernel_counters.json  (the hot kernel's own figures: its name as traced, HBM bytes, VALU issue).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
KT_DIR = {"c3": "c3", "c2": "c2", "c5": "c5", "c3-murray": "c3m", "c3-rosen2fixed": "c3r2f"}


def lpc_for(d):
    nb, l = (d + 3) // 4, 1
    while l < nb:
        l <<= 1
    return l


for cfg in ("c3", "c2", "c5", "c3-murray", "c3-rosen2fixed"):
    bj = os.path.join(go, "%s_bench_%s.json" % (rnd, cfg))
    line = None
    if os.path.exists(bj):
        line = json.load(open(bj))
        json.dump(line, open(os.path.join(pr, "%s_bench_%s.json" % (rnd, cfg)), "w"), indent=1)
    ks = glob.glob(os.path.join(go, "%s_kt_%s" % (rnd, KT_DIR[cfg]), "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(max(ks, key=os.path.getmtime), os.path.join(pr, "%s_%s_kernel_stats.csv" % (rnd, cfg)))  # the latest run's
    table = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(go, "%s_pmc_%s" % (rnd, cfg), "*", "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for (k, c), v in acc.items():
            h = v[len(v) // 2:]
            dd = dur[(k, c)][len(v) // 2:]
            table[k][c] = dict(dispatches_of_measured_job=len(h), mean=sum(h) / len(h), max=max(h),
                               mean_duration_ns_under_profiler=sum(dd) / len(dd))
    if not table:
        continue
    json.dump({"source": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --pmc-child --config %s (one pass per counter "
                         "group, see PMC_PASSES in bench.py), spawned "
                         "by bench.py itself; FETCH_SIZE/WRITE_SIZE in KB" % cfg, "kernels": table},
              open(os.path.join(pr, "%s_%s_pmc_counters.json" % (rnd, cfg)), "w"), indent=1)
    d = line["config"]["nparam"] if line else 16
    # the kernel the bench line names -- the one that ran, as its own child kernel trace called it
    match = ((line or {}).get("roofline") or {}).get("kernel") or "k_fused_fast<%d, true" % lpc_for(d)
    main = [k for k in table if match in k]
    if not main:
        continue
    m = table[main[0]]
    res = {"kernel": main[0], "round": rnd, "config": cfg}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        res["traffic_bytes_per_launch"] = (2 * m["FETCH_SIZE"]["mean"] + m["WRITE_SIZE"]["mean"]) * 1024
        res["traffic_formula"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM)"
    if line and line.get("roofline") and line["roofline"].get("frac") is not None:
        res["valu_busy_fraction"] = line["roofline"]["frac"]  # as bench.py computed it live from these counters
        res["valu_formula"] = "DESIGN.md section 6, 'the bench line': roofline.frac"
        res["SQ_INSTS_VALU"] = m.get("SQ_INSTS_VALU", {}).get("mean")
    json.dump(res, open(os.path.join(pr, "%s_%s_fused_kernel_counters.json" % (rnd, cfg)), "w"), indent=1)
    print(cfg, {k: v for k, v in res.items() if k in ("traffic_bytes_per_launch", "valu_busy_fraction")})

# the strong-scaling shape's own line (tools/refresh_profiles.sh writes it last)
_extra = os.path.join(go, "%s_bench_c3_8192chains.json" % rnd)
if os.path.exists(_extra):
    shutil.copy(_extra, os.path.join(pr, "%s_bench_c3_8192chains.json" % rnd))
