#!/usr/bin/env python3
"""Small-n mode (MCX_OPT_SPLIT_RNG) on/off: whole R-local job (500 burn-in + 1000 main steps, samples kept)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E

def pinit(d, n):
    g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)

for d, n in ((8, 4096), (16, 2048), (16, 8192), (16, 16384), (16, 24576), (16, 32768), (16, 65536), (32, 8192)):
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    res = {}
    for split in (0, 1):
        e = M.Engine(d, n, pl=1.0)
        e.set_option(E.OPT_SPLIT_RNG, split)
        e.stage_pinit(pinit(d, n))
        e.run(1000, 500, None, vl)
        t0 = time.perf_counter()
        for _ in range(5):
            e.run(1000, 500, None, vl)
        res[split] = (time.perf_counter() - t0) / 5
        e.close()
    print("d %2d n %6d (%5d waves): fused %.3f ms  split %.3f ms  -> %.2fx   %.3e / %.3e chain-steps/s"
          % (d, n, n * max(1, d // 4) // 64, res[0] * 1e3, res[1] * 1e3, res[0] / res[1], n * 1500 / res[0], n * 1500 / res[1]))
