#!/usr/bin/env python3
"""Pricing a per-PAIR screen for the dense Murray sweeps of C5 before writing one.  After 10, 30, 50,
70 and 100 main-loop steps of the C5 per-GPU job (32-D 8-component mixture x 32 768 chains, pl 0.9): which share of the
(chain, Q_i) pairs has arg <= 176 -- the only ones whose Q_i is not exactly 0 --, and for which share of the
(group of G neighbouring chains, Q_i) ROWS every chain of the group is past 176, the unit a lock-step wavefront can skip
(G = 128: what a wavefront of the sweep holds; 16; 4), with the chains in index order and sorted along the mixture's axis.
Measurement helper; runs on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcpar_amd as M  # noqa: E402
from bench import mix_params, pinit_for  # noqa: E402

d, K, n = 32, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 32768
par = mix_params(d, K)
for nsamp in (10, 30, 50, 70, 100):
    eng = M.Engine(d, n, pl=0.9)
    vl, keep = M.make_vlfunc(M.VL_GAUSSMIX, d, par, K)
    eng.run(nsamp, 500, pinit_for(d, n, 0), vl)
    x = eng.state.astype(np.float32)
    ms = eng.musigall.reshape(n, d, 2)
    mu, w = ms[:, :, 0], (1.0 / ms[:, :, 1]).astype(np.float32)
    order = np.argsort(x.sum(axis=1), kind="stable")  # position along the line the eight means lie on
    rng = np.random.default_rng(1)
    qi = rng.choice(n, 2048, replace=False)           # a sample of the Gaussians
    out = {}
    for name, perm in (("index order", np.arange(n)), ("sorted along the mixture axis", order)):
        xs = x[perm]
        dead_rows = {128: 0, 16: 0, 4: 0}
        alive = 0
        for q0 in range(0, len(qi), 64):
            q = qi[q0:q0 + 64]
            arg = np.zeros((n, len(q)), np.float32)
            for k in range(d):  # (ascending k like the sweep; float32)
                t = mu[q, k][None, :] - xs[:, k][:, None]
                arg += t * t * w[q, k][None, :]
            a = arg <= 176.0
            alive += int(a.sum())
            for G in dead_rows:
                dead_rows[G] += int((~a.reshape(n // G, G, len(q)).any(axis=1)).sum())
        tot = n * len(qi)
        out[name] = (alive / tot, {G: v / (tot / G) for G, v in dead_rows.items()})
    c = eng.counters
    print("after %3d main steps (%d remote steps, %d passes): pairs with arg <= 176: %.4f" % (nsamp, c["remote_steps"], c["remote_passes"], out["index order"][0]))
    for name, (al, dr) in out.items():
        print("     rows a lock-step group could skip, chains in %s: G=128 %.4f   G=16 %.4f   G=4 %.4f" % (name, dr[128], dr[16], dr[4]))
    eng.close()
