#!/usr/bin/env python3
"""Where a small-n job's wall time goes on the host (MCX_VERBOSE=2: set-up, queueing, waiting for the stream) next to the
job time; samples kept / not kept.  usage: host_timing_probe.py [d n]"""
import os
import sys
import time

os.environ["MCX_VERBOSE"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mcpar_amd as M  # noqa: E402
from mcpar_amd import engine as E  # noqa: E402
from persist_sweep import pinit  # noqa: E402

d, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 8192)
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
for samples in (1, 0):
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_SAMPLES, samples)
    e.stage_pinit(pinit(d, n))
    for nburn, nsamp in ((500, 1000), (0, 1000), (0, 2000)):
        for _ in range(3):
            e.run(nsamp, nburn, None, vl)
        sys.stderr.write("--- samples=%d nburn=%d nsamp=%d\n" % (samples, nburn, nsamp))
        sys.stderr.flush()
        t0 = time.perf_counter()
        for _ in range(5):
            e.run(nsamp, nburn, None, vl)
        sys.stderr.write("--- job %.1f us\n" % ((time.perf_counter() - t0) / 5 * 1e6))
    e.close()
