#!/usr/bin/env python3
"""Whole-job time (500 burn-in + 1000 main steps, samples kept) of the one-launch small-n mode against the fused
kernels, by chain count: where MCX_OPT_PERSIST's automatic choice should switch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E

def pinit(d, n):
    g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)

def job_ms(kind, d, n, persist, params=None, K=0, reps=20):
    vl, keep = M.make_vlfunc(kind, d, params, K)
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_PERSIST, persist)
    if not persist:
        e.set_option(E.OPT_SPLIT_RNG, 0)
    e.stage_pinit(pinit(d, n))
    for _ in range(3):
        e.run(1000, 500, None, vl)
    t0 = time.perf_counter()
    for _ in range(reps):
        e.run(1000, 500, None, vl)
    dt = (time.perf_counter() - t0) / reps
    launches = e.counters["kernel_launches"]
    e.close()
    return dt * 1e3, launches

if __name__ == "__main__":
  for d in (8, 16, 32):
      for n in (8192, 16384, 20480, 24576, 28672, 32768, 49152, 65536, 131072):
          lpc = 1
          while lpc * 4 < d: lpc *= 2
          if n * lpc // 64 > 8 * 256:
              continue
          a, la = job_ms(M.VL_ROSENBROCK1, d, n, 1)
          b, lb = job_ms(M.VL_ROSENBROCK1, d, n, 0)
          print("d=%2d n=%6d owners/WG=%d  one launch %.3f ms (%d launches)   fused %.3f ms (%d launches)   ratio %.2f" % (d, n, -(-(n * lpc // 64) // 256), a, la, b, lb, b / a), flush=True)
