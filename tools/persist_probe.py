#!/usr/bin/env python3
"""What the one-launch kernel's step time depends on: the likelihood's cost (Rosenbrock1 against a diagonal Gaussian), the
lanes per chain (the depth of the lane-group reduction) and the owner wavefronts per workgroup, at equal wavefront counts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import mcpar_amd as M  # noqa: E402
from persist_sweep import job_ms  # noqa: E402

for d, n in ((4, 16384), (8, 8192), (16, 4096), (32, 2048), (4, 32768), (8, 16384), (16, 8192), (32, 4096)):
    lpc = 1
    while lpc * 4 < d:
        lpc *= 2
    g = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
    a, _ = job_ms(M.VL_ROSENBROCK1, d, n, 1)
    b, _ = job_ms(M.VL_GAUSSIAN, d, n, 1, params=g)
    print("d=%2d n=%6d lanes/chain=%d owner wavefronts=%4d   Rosenbrock1 %.3f ms   Gaussian %.3f ms" % (d, n, lpc, n * lpc // 64, a, b), flush=True)
