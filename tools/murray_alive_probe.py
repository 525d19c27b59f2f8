#!/usr/bin/env python3
"""How sparse is a Murray sum sweep, pair by pair?  After a C3 R-murray job: the fraction of (chain, Q_i) pairs whose
partial arg is still <= 176 after 4, 8, 12 and all 16 dimensions (what a per-lane -- instead of per-wavefront --
early-out could drop).  Measurement helper; runs on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mcpar_amd as M  # noqa: E402
import oracle_lib as O  # noqa: E402

d, n = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = M.Engine(d, n, pl=0.9)
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
eng.run(100, 500, O.default_pinit(d, n), vl)
x = eng.state.astype(np.float32)
ms = eng.musigall.reshape(n, d, 2)
mu, w = ms[:, :, 0], (1.0 / ms[:, :, 1]).astype(np.float32)
rng = np.random.default_rng(1)
pick = rng.choice(n, 512, replace=False)
alive = np.zeros(4)
tot = 0
for c0 in range(0, len(pick), 32):
    xs = x[pick[c0:c0 + 32]]                      # [32, d]
    t = (mu[None, :, :] - xs[:, None, :]) ** 2 * w[None, :, :]   # [32, n, d]
    cs = np.cumsum(t, axis=2)
    for gi, k in enumerate((3, 7, 11, 15)):
        alive[gi] += np.count_nonzero(cs[:, :, k] <= 176.0)
    tot += t.shape[0] * t.shape[1]
print("pairs alive after 4 / 8 / 12 / 16 dimensions:", " ".join("%.5f" % (a / tot) for a in alive))
print("counters:", {k: eng.counters[k] for k in ("remote_steps", "remote_passes", "remote_pairs_evaluated")})
