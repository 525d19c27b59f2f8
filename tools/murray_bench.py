#!/usr/bin/env python3
"""R-murray runs of SURVEY 8d (pl = 0.9, nburn = 500, nsamp = 100) on one GPU, pass counts reported."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E

def pinit(d, n):
    g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)

def mix(d, K):
    means = np.stack([np.full(d, 5.0 * k / (K - 1)) for k in range(K)]).astype(np.float32)
    w = np.ones(K, np.float32); w[0] = 5
    return np.concatenate([means.ravel(), w])

for name, kind, d, n, params, K in (("C2 rosen1 8-D x 4096", M.VL_ROSENBROCK1, 8, 4096, None, 0),
                                    ("rosen1 16-D x 16384", M.VL_ROSENBROCK1, 16, 16384, None, 0),
                                    ("C3 rosen1 16-D x 65536", M.VL_ROSENBROCK1, 16, 65536, None, 0),
                                    ("C5/GPU mix 32-D K=8 x 32768", M.VL_GAUSSMIX, 32, 32768, mix(32, 8), 8)):
    vl, keep = M.make_vlfunc(kind, d, params, K)
    e = M.Engine(d, n, pl=0.9)
    e.set_option(E.OPT_SAMPLES, 0)
    e.set_option(E.OPT_PROFILE, 1)
    p = pinit(d, n)
    t0 = time.perf_counter()
    e.run(100, 500, p, vl)
    dt = time.perf_counter() - t0
    c = e.counters; pr = e.profile
    print("%-30s %.3f s  %.3e chain-steps/s  accept %.3f  remote steps %d passes %d  remote kernels %.1f ms (%.2f ms/pass)"
          % (name, dt, n * 600 / dt, c["naccept_main"] / (n * 100.0), c["remote_steps"], c["remote_passes"],
             pr["remote"]["ms"], pr["remote"]["ms"] / max(1, c["remote_passes"])))
    e.close()
