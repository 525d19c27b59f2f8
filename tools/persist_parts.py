#!/usr/bin/env python3
"""One-launch small-n kernel: where a job's time goes -- burn-in steps against main-loop steps, with and without the
sample store -- by blocks per lane.  usage: persist_parts.py [d n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mcpar_amd as M  # noqa: E402
from mcpar_amd import engine as E  # noqa: E402
from persist_sweep import pinit  # noqa: E402

d, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 8192)
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
for bpl in (1, 2):
    if d % (4 * bpl):
        continue
    for samples in (1, 0):
        for nburn, nsamp in ((500, 1000), (1500, 1), (500, 1), (0, 1000), (0, 2000)):
            e = M.Engine(d, n, pl=1.0)
            e.set_option(E.OPT_PERSIST, 1)
            e.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
            e.set_option(E.OPT_SAMPLES, samples)
            e.stage_pinit(pinit(d, n))
            for _ in range(4):
                e.run(nsamp, nburn, None, vl)
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(20):
                    e.run(nsamp, nburn, None, vl)
                best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
            c = e.counters
            print("d=%d n=%d bpl=%d samples=%d nburn=%4d nsamp=%4d: %.4f ms  (%.1f ns/step, %d launches, bpl %d)"
                  % (d, n, bpl, samples, nburn, nsamp, best, best * 1e6 / (nburn + nsamp), c["kernel_launches"], c["small_n_blocks_per_lane"]), flush=True)
            e.close()
