#!/usr/bin/env python3
"""Launch-duration of the fused main-loop kernel vs steps per launch (HIP events per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E

d, n = 16, 65536
g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
p = (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
for samples in (1, 0):
    for seg in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        e = M.Engine(d, n, pl=1.0)
        e.set_option(E.OPT_SAMPLES, samples)
        e.set_option(E.OPT_MAX_SEGMENT, seg)
        e.run(seg * 8, 0, p, vl)
        e.set_option(E.OPT_PROFILE, 1)
        b = e.profile
        e.run(seg * 16, 0, p, vl)
        pr = e.profile["fused_main"]
        ms = (pr["ms"] - b["fused_main"]["ms"]) / (pr["launches"] - b["fused_main"]["launches"])
        print("samples %d  steps/launch %4d  launch %.1f us  per step %.2f us" % (samples, seg, ms * 1e3, ms * 1e3 / seg))
        e.close()
