#!/usr/bin/env python3
"""Pricing the matrix-core screen of the Murray sweeps before writing it: at a few points of the C5 and C3-murray jobs,
which share of the (group of 128 chains, Q_i) rows could a sweep skip
  * at best (every chain of the group truly past its bound),
  * with the bf16 screen: arg >= c_i + A_j . B_i - 2^-6.8 |A_j| |B_i|  (A_j = (y, y^2), B_i = (-2 w m, w), both
    rounded to bf16, y and m centred on the Gaussians' mean; Cauchy-Schwarz bounds the rounding of the products),
for the sum sweeps (bound 176) and the min-arg sweep (bound min(176, the chain's arg against its own Gaussian)), with the
chains in index order and in the Z-order of the box screen's sort (approximated here by a sort on four coordinates).
Measurement helper; runs on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcpar_amd as M  # noqa: E402
from bench import CONFIGS, mix_params, pinit_for  # noqa: E402


def bf16(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def zorder(x):
    d = x.shape[1]
    dims = [c * (d // 4) for c in range(4)]
    key = np.zeros(len(x), np.int64)
    q = []
    for k in dims:
        v = x[:, k]
        lo, hi = v.mean() - 2 * v.std(), v.mean() + 2 * v.std()
        q.append(np.clip(((v - lo) / (hi - lo + 1e-30) * 8).astype(np.int64), 0, 7))
    for b in range(2, -1, -1):
        for c in range(4):
            key = (key << 1) | ((q[c] >> b) & 1)
    return np.argsort(key, kind="stable")


def probe(cfgname, points, nq):
    cfg = CONFIGS[cfgname]
    d, n = cfg["d"], cfg["n"]
    for nsamp in points:
        eng = M.Engine(d, n, pl=cfg["pl"])
        if cfg["lik"] == 5:
            vl, keep = M.make_vlfunc(M.VL_GAUSSMIX, d, mix_params(d, cfg["K"]), cfg["K"])
        else:
            vl, keep = M.make_vlfunc(cfg["lik"], d)
        eng.run(nsamp, cfg["nburn"], pinit_for(d, n, 0), vl)
        x = eng.state.astype(np.float32)
        ms = eng.musigall.reshape(n, d, 2)
        mu, w = ms[:, :, 0].astype(np.float32), (1.0 / ms[:, :, 1]).astype(np.float32)
        own = ((mu - x) ** 2 * w).sum(axis=1)
        cen = mu.mean(axis=0)
        y, m = x - cen, mu - cen
        A = np.concatenate([y, y * y], axis=1)
        B = np.concatenate([-2 * w * m, w], axis=1)
        cst = (w.astype(np.float64) * m.astype(np.float64) ** 2).sum(axis=1)
        na, nb = np.linalg.norm(A.astype(np.float64), axis=1), np.linalg.norm(B.astype(np.float64), axis=1)
        Ab, Bb = bf16(A).astype(np.float64), bf16(B).astype(np.float64)
        rng = np.random.default_rng(1)
        qi = rng.choice(n, nq, replace=False)
        c = eng.counters
        print("%s after %3d main steps (%d remote steps, %d passes)" % (cfgname, nsamp, c["remote_steps"], c["remote_passes"]), flush=True)
        for oname, perm in (("index order", np.arange(n)), ("z-order", zorder(x))):
            res = {}
            for q0 in range(0, nq, 256):
                q = qi[q0:q0 + 256]
                arg = np.zeros((n, len(q)), np.float32)
                for k in range(d):
                    t = mu[q, k][None, :] - x[perm, k][:, None]
                    arg += t * t * w[q, k][None, :]
                low = cst[q][None, :] + Ab[perm] @ Bb[q].T - 2.0 ** -6.8 * na[perm][:, None] * nb[q][None, :]
                for kind, lim in (("sums", np.full(n, 176.0)), ("min-arg", np.minimum(176.0, own[perm]))):
                    alive = arg <= lim[:, None]
                    salive = ~(low > lim[:, None])
                    assert not (alive & ~salive).any(), "the screen dropped a live pair"
                    r = res.setdefault(kind, [0, 0, 0, 0])
                    r[0] += int(alive.sum())
                    r[1] += int((~alive.reshape(n // 128, 128, -1).any(axis=1)).sum())
                    r[2] += int((~salive.reshape(n // 128, 128, -1).any(axis=1)).sum())
                    r[3] += (n // 128) * len(q)
            for kind, r in res.items():
                print("   %-8s %-12s pairs alive %.5f   rows of 128 skippable: at best %.4f   by the bf16 screen %.4f"
                      % (kind, oname, r[0] / (n * nq), r[1] / r[3], r[2] / r[3]), flush=True)
        eng.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("c5", "both"):
        probe("c5", (10, 40, 70, 100), 1024)
    if which in ("c3", "both"):
        probe("c3-murray", (10, 50, 100), 512)
