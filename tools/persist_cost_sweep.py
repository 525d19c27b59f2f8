#!/usr/bin/env python3
"""One-launch small-n kernel: job time against the cost model of the generator deal (MCX_PERSIST_COST = cn,ca,co,cr[,map];
mcx_k_persist.hip, mcxk_persist_deal).  One process per point.  usage: persist_cost_sweep.py d n bpl [d n bpl]..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from persist_bpl_probe import one  # noqa: E402


def point(d, n, bpl, cost, k=None):
    os.environ["MCX_PERSIST_COST"] = cost
    r = one(d, n, bpl, k, None)
    print("d=%d n=%d bpl=%d K=%s cost=%-22s -> %s" % (d, n, bpl, k, cost, r), flush=True)
    return r.get("ms", 9e9)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    shapes = [(a[i], a[i + 1], a[i + 2]) for i in range(0, len(a) - 2, 3)] or [(16, 8192, 2), (16, 8192, 1), (8, 4096, 1)]
    for d, n, bpl in shapes:
        best = (9e9, None)
        for co in (0, 20, 35, 50, 70, 100, 150, 1000):
            for cr in (0, 25, 40, 60, 100):
                c = "128,110,%d,%d" % (co, cr)
                ms = point(d, n, bpl, c)
                best = min(best, (ms, c))
        print("best for d=%d n=%d bpl=%d: %s" % (d, n, bpl, best), flush=True)
        # the other SIMD mapping (wavefront w on SIMD w / 4) with the best costs: slower if w % 4 is the truth
        point(d, n, bpl, best[1] + ",1")
