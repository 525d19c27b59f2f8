#!/usr/bin/env python3
"""8192 x 16-D jobs queued back to back (MCX_OPT_ASYNC_RUN): wall per job, and -- under `rocprofv3 --kernel-trace` with
--trace-dir -- the kernel's own duration against the start-to-start interval (what the queue adds between two jobs).
usage: queued_jobs_probe.py [--jobs 200] [--sync]      |      queued_jobs_probe.py --read DIR"""
import argparse
import csv
import glob
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
    ev = [e for e in ev if "k_run_small" in e[2]]
    ev = ev[len(ev) // 4:]
    dur = [e - s for s, e, _ in ev]
    gap = [b[0] - a[1] for a, b in zip(ev, ev[1:])]
    print("%d launches of k_run_small: duration median %.1f us (p10 %.1f, p90 %.1f); idle before the next one median %.1f us (p10 %.1f, p90 %.1f)"
          % (len(ev), statistics.median(dur) / 1e3, sorted(dur)[len(dur) // 10] / 1e3, sorted(dur)[9 * len(dur) // 10] / 1e3,
             statistics.median(gap) / 1e3, sorted(gap)[len(gap) // 10] / 1e3, sorted(gap)[9 * len(gap) // 10] / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=200)
    ap.add_argument("--sync", action="store_true")
    ap.add_argument("--by-copy", action="store_true", help="MCX_OPT_SELF_REPORT = 0: the counters by a copy behind the kernel")
    ap.add_argument("--read")
    a = ap.parse_args()
    if a.read:
        return read(a.read)
    import numpy as np
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 16, 8192, 500, 1000
    g = np.arange(n * d, dtype=np.float64)
    p = (0.5 * np.sin(0.37 * g)).astype(np.float32).reshape(n, d)
    vl, _k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng = M.Engine(d, n, pl=1.0)
    eng.stage_pinit(p)
    eng.set_option(E.OPT_ASYNC_RUN, 0 if a.sync else 1)
    eng.set_option(E.OPT_SELF_REPORT, 0 if a.by_copy else 1)
    for _ in range(6):
        eng.run(nsamp, nburn, None, vl)
    eng.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(a.jobs):
            eng.run(nsamp, nburn, None, vl)
        eng.synchronize()
        print("%s%s: %.1f us per job over %d jobs" % ("one by one" if a.sync else "queued", ", counters by copy" if a.by_copy else "", (time.perf_counter() - t0) / a.jobs * 1e6, a.jobs))
    print("meet_timeouts_total", eng.counters["meet_timeouts_total"])
    eng.close()


if __name__ == "__main__":
    main()
