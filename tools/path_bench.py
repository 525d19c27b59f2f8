#!/usr/bin/env python3
"""Per-step cost of the different kernel paths on 65 536 chains (main-loop fused kernel, HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E

def pinit(d, n):
    g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)

def run(label, kind, d, n, params=None, K=0, incov=None, mask=0, fuse=1):
    vl, keep = M.make_vlfunc(kind, d, params, K)
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_ACCEPT_MASK, mask)
    e.set_option(E.OPT_FUSE, fuse)
    p = pinit(d, n)
    e.run(200, 100, p, vl, incov)
    e.set_option(E.OPT_PROFILE, 1)
    b = e.profile
    e.run(200, 100, p, vl, incov)
    pr = e.profile
    if fuse:
        ms = pr["fused_main"]["ms"] - b["fused_main"]["ms"]
    else:
        ms = sum(pr[k]["ms"] - b[k]["ms"] for k in ("propose", "eval", "accept")) * 2.0 / 3.0  # main = 200 of 300 steps
    print("%-44s %.2f us/step  %.2e chain-steps/s" % (label, ms * 1e3 / 200, n * 200 / (ms * 1e-3)))
    e.close()

n = 65536
rng = np.random.default_rng(0)
def spd(d):
    a = rng.normal(size=(d, d)).astype(np.float32)
    return (a @ a.T / d + 0.5 * np.eye(d)).astype(np.float32)
run("rosen1 d16 fast kernel", M.VL_ROSENBROCK1, 16, n)
run("rosen1 d16 generic kernel (mask on)", M.VL_ROSENBROCK1, 16, n, mask=1)
run("rosen1 d16 full covariance", M.VL_ROSENBROCK1, 16, n, incov=spd(16))
run("rosen1 d16 unfused (3 kernels/step)", M.VL_ROSENBROCK1, 16, n, fuse=0)
run("gauss d16 fast kernel", M.VL_GAUSSIAN, 16, n)
run("rosen1 d8 fast", M.VL_ROSENBROCK1, 8, n)
run("rosen1 d32 fast", M.VL_ROSENBROCK1, 32, n)
run("rosen1 d12 (ragged lanes, fast)", M.VL_ROSENBROCK1, 12, n)
run("rosen1 d14 (generic, scalar loads)", M.VL_ROSENBROCK1, 14, n)
run("rosen1 d64 generic", M.VL_ROSENBROCK1, 64, n)
run("rosen1 d32 full covariance", M.VL_ROSENBROCK1, 32, n, incov=spd(32))
means = np.stack([np.full(32, 5.0 * k / 7) for k in range(8)]).astype(np.float32)
run("mix d32 K8", M.VL_GAUSSMIX, 32, 32768, np.concatenate([means.ravel(), [5, 1, 1, 1, 1, 1, 1, 1]]).astype(np.float32), 8)
