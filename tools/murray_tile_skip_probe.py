#!/usr/bin/env python3
"""Would sorting the GAUSSIANS too let the matrix-core screen skip whole tiles?  At a few points of the C3-murray and C5
jobs: chains Z-order-sorted into groups of 128, Gaussians Z-order-sorted into tiles of 32 (by their means, same four key
coordinates); a (group, tile) pair is skippable without any product when the boxes of the two are farther apart, in the
key coordinates and with the tile's smallest weights, than the group's bound (176).  Prints the share of such pairs --
the share of the screen's matrix instructions that would not have to be issued.  Measurement helper; GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcpar_amd as M  # noqa: E402
from bench import CONFIGS, mix_params, pinit_for  # noqa: E402
from murray_gemm_screen_probe import zorder  # noqa: E402


def probe(cfgname, points):
    cfg = CONFIGS[cfgname]
    d, n = cfg["d"], cfg["n"]
    dims = [c * (d // 4) for c in range(4)]
    for nsamp in points:
        eng = M.Engine(d, n, pl=cfg["pl"])
        if cfg["lik"] == 5:
            vl, keep = M.make_vlfunc(M.VL_GAUSSMIX, d, mix_params(d, cfg["K"]), cfg["K"])
        else:
            vl, keep = M.make_vlfunc(cfg["lik"], d)
        eng.run(nsamp, cfg["nburn"], pinit_for(d, n, 0), vl)
        x = eng.state.astype(np.float32)
        ms = eng.musigall.reshape(n, d, 2)
        mu, w = ms[:, :, 0].astype(np.float64), 1.0 / ms[:, :, 1].astype(np.float64)
        xs = x[zorder(x)][:, dims].astype(np.float64)
        go = zorder(mu.astype(np.float32))
        mus, ws = mu[go][:, dims], w[go][:, dims]
        for tile in (32, 64):
            glo, ghi = xs.reshape(n // 128, 128, 4).min(axis=1), xs.reshape(n // 128, 128, 4).max(axis=1)
            tlo, thi = mus.reshape(n // tile, tile, 4).min(axis=1), mus.reshape(n // tile, tile, 4).max(axis=1)
            twm = ws.reshape(n // tile, tile, 4).min(axis=1)
            dead = 0
            for g0 in range(0, n // 128, 64):
                gap = np.maximum(np.maximum(tlo[None, :, :] - ghi[g0:g0 + 64, None, :], glo[g0:g0 + 64, None, :] - thi[None, :, :]), 0.0)
                dead += int(((gap * gap * twm[None, :, :]).sum(axis=2) > 176.0).sum())
            print("%s after %3d main steps: (group of 128 chains, tile of %d Gaussians) pairs skippable by boxes alone: %.4f"
                  % (cfgname, nsamp, tile, dead / float((n // 128) * (n // tile))), flush=True)
        eng.close()


if __name__ == "__main__":
    probe("c3-murray", (10, 50, 100))
    probe("c5", (10, 40, 70))
