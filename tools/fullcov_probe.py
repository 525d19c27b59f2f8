#!/usr/bin/env python3
"""Where the time of a full-covariance job goes: fused burn-in / main-loop kernels, diagonal against full factor,
with and without sample rows (HIP events, MCX_OPT_PROFILE)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E
from bench import pinit_for, spd_covariance

BPL = int(os.environ.get("MCX_PROBE_BPL", "0"))  # MCX_OPT_BLOCKS_PER_LANE: 0 = the engine's choice, 1 = one block per lane, 2 = two (full covariance: mirrored)


def probe(d, full, samples, nburn=500, nsamp=1000, n=65536):
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_SAMPLES, samples)
    e.set_option(E.OPT_BLOCKS_PER_LANE, BPL)
    p = pinit_for(d, n, 0)
    cov = spd_covariance(d) if full else None
    e.run(nsamp, nburn, p, vl, cov)
    t0 = time.perf_counter()
    e.run(nsamp, nburn, p, vl, cov)
    wall = time.perf_counter() - t0
    e.set_option(E.OPT_PROFILE, 1)
    b = e.profile
    e.run(nsamp, nburn, p, vl, cov)
    pr = e.profile
    c = e.counters
    get = lambda k: pr[k]["ms"] - b[k]["ms"]
    print("d=%d %s samples=%d: wall %.2f ms | burn %.2f ms main %.2f ms tuner %.3f misc %.3f | launches %d | accept %.3f trace %s"
          % (d, "full" if full else "diag", samples, wall * 1e3, get("fused_burn"), get("fused_main"), get("tuner"), get("misc"),
             c["kernel_launches"], c["naccept_main"] / (n * nsamp), e.tuner_trace[:10]))
    e.close()

for d in (16, 32):
    for full in (0, 1):
        for samples in (0, 1) if d == 16 else (0,):
            probe(d, full, samples)

# bench.py's order: ONE engine, staged pinit, diagonal job first, then the full-covariance job; every repetition timed
print("-- one engine, diagonal then full (bench.py claims_under_the_clock order)")
for d in (16, 32):
    n = 65536
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_SAMPLES, 0)
    e.set_option(E.OPT_BLOCKS_PER_LANE, BPL)
    e.stage_pinit(pinit_for(d, n, 0))
    for label, cov in (("diag", None), ("full", spd_covariance(d)), ("diag again", None), ("full again", spd_covariance(d))):
        ts = []
        for r in range(5):
            t0 = time.perf_counter()
            e.run(1000, 500, None, vl, cov)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("d=%d %-10s reps ms: %s" % (d, label, " ".join("%.2f" % t for t in ts)))
    e.close()
