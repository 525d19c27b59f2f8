#!/usr/bin/env python3
"""MCX_OPT_MURRAY_OVERLAP A/B on one box: C3-murray and C5 (per-GPU shape) jobs with the big passes' screen and sweep
alternating (0) and by column chunks on two streams (2, 4, 8, 16), ms per job (median of 9 after 3 warm runs) and the bits
of the final state against the unchunked run.  usage: murray_overlap_ab.py [chunks ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcpar_amd as M  # noqa: E402
from mcpar_amd import engine as E  # noqa: E402


def pinit(d, n):
    g = np.arange(n, dtype=np.float64)[:, None]
    i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)


def mix(d, K):
    means = np.stack([np.full(d, 5.0 * k / (K - 1)) for k in range(K)]).astype(np.float32)
    w = np.ones(K, np.float32)
    w[0] = 5
    return np.concatenate([means.ravel(), w])


quick = "--trace" in sys.argv  # under rocprofv3: C3-murray only, three jobs per setting
chunks = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [0, 2, 4, 8, 16]
shapes = (("C3-murray 16-D x 65536", M.VL_ROSENBROCK1, 16, 65536, None, 0),
          ("C5/GPU mix 32-D K=8 x 32768", M.VL_GAUSSMIX, 32, 32768, mix(32, 8), 8))
for name, kind, d, n, params, K in (shapes[:1] if quick else shapes):
    vl, keep = M.make_vlfunc(kind, d, params, K)
    ref = None
    for c in chunks:
        e = M.Engine(d, n, pl=0.9)
        e.set_option(E.OPT_MURRAY_OVERLAP, c)
        e.stage_pinit(pinit(d, n))
        ts = []
        for r in range(5 if quick else 12):
            t0 = time.perf_counter()
            e.run(100, 500, None, vl)
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2 if quick else 3:])
        st = e.state.view(np.uint32).copy()
        cn = e.counters
        if ref is None:
            ref = (st, cn["remote_passes"])
        same = bool(np.array_equal(st, ref[0])) and cn["remote_passes"] == ref[1]
        print("%-30s chunks %2d: %.3f ms per job (min %.3f)  passes %d  same bits as the first: %s"
              % (name, c, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, cn["remote_passes"], same), flush=True)
        e.close()
