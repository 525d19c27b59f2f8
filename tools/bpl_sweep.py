#!/usr/bin/env python3
"""Job time of the hot-path kernel with 1, 2 and 4 blocks per lane (MCX_OPT_BLOCKS_PER_LANE)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E
from bench import pinit_for, mix_params


def job(d, n, bpl, lik="rosen", nburn=500, nsamp=1000, samples=1, reps=9):
    if lik == "rosen":
        vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    else:
        vl, keep = M.make_vlfunc(M.VL_GAUSSMIX, d, mix_params(d, 8), 8)
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
    e.set_option(E.OPT_SAMPLES, samples)
    e.set_option(E.OPT_PERSIST, 0)
    e.set_option(E.OPT_SPLIT_RNG, 0)
    e.stage_pinit(pinit_for(d, n, 0))
    ts = []
    for r in range(reps + 2):
        t0 = time.perf_counter()
        e.run(nsamp, nburn, None, vl)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    e.set_option(E.OPT_PROFILE, 1)
    b = e.profile
    e.run(nsamp, nburn, None, vl)
    pr = e.profile
    km = pr["fused_main"]["ms"] - b["fused_main"]["ms"]
    kb = pr["fused_burn"]["ms"] - b["fused_burn"]["ms"]
    e.close()
    t = ts[len(ts) // 2]
    print("%-6s d=%-3d n=%-7d bpl=%d samples=%d: job %.3f ms (min %.3f)  %.3e chain-steps/s | kernels: burn %.3f main %.3f ms"
          % (lik, d, n, bpl, samples, t * 1e3, ts[0] * 1e3, n * (nburn + nsamp) / t, kb, km), flush=True)


for bpl in (1, 2, 4):
    job(16, 65536, bpl)
for bpl in (1, 2, 4):
    job(16, 196608, bpl, samples=0)
for bpl in (1, 2, 4):
    job(16, 32768, bpl)
for bpl in (1, 2):
    job(8, 131072, bpl)
for bpl in (1, 2, 4):
    job(32, 32768, bpl, lik="mix", nsamp=100, samples=0)
for bpl in (1, 2, 4):
    job(32, 65536, bpl, samples=0)
