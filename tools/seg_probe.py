#!/usr/bin/env python3
"""Why are 10-step launches slow inside a long job?  Vary one thing at a time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E

d, n = 16, 65536
g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
p = (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)

def probe(label, nburn, nsamp, seg, samples):
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_SAMPLES, samples)
    e.set_option(E.OPT_MAX_SEGMENT, seg)
    e.run(nsamp, nburn, p, vl)
    e.set_option(E.OPT_PROFILE, 1)
    b = e.profile
    e.run(nsamp, nburn, p, vl)
    pr = e.profile
    fm = pr["fused_main"]; fb = pr["fused_burn"]
    ms = (fm["ms"] - b["fused_main"]["ms"]) / max(1, fm["launches"] - b["fused_main"]["launches"])
    mb = (fb["ms"] - b["fused_burn"]["ms"]) / max(1, fb["launches"] - b["fused_burn"]["launches"])
    print("%-46s main %.1f us/launch (%.2f us/step)   burn %.1f us/launch  acc %.3f" % (
        label, ms * 1e3, ms * 1e3 / seg, mb * 1e3, e.counters["naccept_main"] / (n * nsamp)))
    e.close()

probe("nburn 0   nsamp 160  seg 10 samples", 0, 160, 10, 1)
probe("nburn 500 nsamp 160  seg 10 samples", 500, 160, 10, 1)
probe("nburn 500 nsamp 1000 seg 10 samples", 500, 1000, 10, 1)
probe("nburn 500 nsamp 1000 seg 10 no samples", 500, 1000, 10, 0)
probe("nburn 500 nsamp 1000 seg 50 samples", 500, 1000, 50, 1)
probe("nburn 0   nsamp 1000 seg 10 samples", 0, 1000, 10, 1)
