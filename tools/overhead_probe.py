#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E
d, n = 16, 65536
g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
p = (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
e = M.Engine(d, n, pl=1.0)
e.stage_pinit(p)
def t(label, nsamp, nburn, reps=20, pin=None):
    e.run(nsamp, nburn, pin, vl)
    t0 = time.perf_counter()
    for _ in range(reps): e.run(nsamp, nburn, pin, vl)
    print("%-28s %.1f us per run" % (label, (time.perf_counter() - t0) / reps * 1e6))
t("nsamp 0 nburn 0", 0, 0)
t("nsamp 1 nburn 0", 1, 0)
t("nsamp 0 nburn 52", 0, 52)
t("nsamp 0 nburn 500", 0, 500)
t("nsamp 1000 nburn 0", 1000, 0)
t("nsamp 1000 nburn 500", 1000, 500)
t("nsamp 1000 nburn 500 hostp", 1000, 500, pin=p)
e.set_option(E.OPT_SAMPLES, 0)
t("nsamp 1000 nburn 500 nosamp", 1000, 500)
