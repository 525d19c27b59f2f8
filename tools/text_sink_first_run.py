#!/usr/bin/env python3
"""The whole C3 job through the text sink, the engine's FIRST run and its next two (what a one-shot driver pays against
what the bench's warm engine pays).  usage: [MCX_LIBMCX=other.so] python tools/text_sink_first_run.py [block_steps]
Measurement helper; GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcpar_amd as M  # noqa: E402
from bench import pinit_for  # noqa: E402

d, n, nburn, nsamp = 16, 65536, 500, 1000
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 25
M.load().mcx_set_device(0)
t0 = time.perf_counter()
eng = M.Engine(d, n, pl=1.0)
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
nbytes = [0]


def sink(first, nsteps, view):
    nbytes[0] += len(view)
    return 0


eng.set_text_sink(sink, blk)
p = pinit_for(d, n, 0)
print("engine made in %.0f ms" % ((time.perf_counter() - t0) * 1e3))
for r in range(3):
    nbytes[0] = 0
    t = time.perf_counter()
    eng.run(nsamp, nburn, p, vl)
    eng.synchronize()
    print("run %d: %.1f ms, %.2f GB of text" % (r, (time.perf_counter() - t) * 1e3, nbytes[0] / 1e9))
eng.close()
