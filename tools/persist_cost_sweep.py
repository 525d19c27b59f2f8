#!/usr/bin/env python3
"""One-launch small-n kernel: job time against the generator deal's cost model (MCX_PERSIST_COST =
cn,ca,co,cr,map,singles; mcx_k_persist.hip, mcxk_persist_deal), recorders yes/no and steps per phase.  One process per
point.  usage: persist_cost_sweep.py d n [quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from persist_bpl_probe import one  # noqa: E402


def point(d, n, bpl, cost, k, rec):
    os.environ["MCX_PERSIST_COST"] = cost
    r = one(d, n, bpl, k, rec)
    print("d=%d n=%d bpl=%d rec=%s K=%s cost=%-24s -> %s" % (d, n, bpl, rec, k, cost, r), flush=True)
    return r.get("ms", 9e9)


if __name__ == "__main__":
    d, n = int(sys.argv[1]), int(sys.argv[2])
    best = (9e9, None)
    for bpl in (1, 2):
        if d % (8 if bpl == 2 else 4):
            continue
        for rec in (1, 0):
            for k in (16, 24, 32):
                for singles in (1, 0):
                    for co, cr in ((34, 24), (50, 40), (80, 50), (20, 10)):
                        c = "185,110,%d,%d,0,%d" % (co, cr, singles)
                        ms = point(d, n, bpl, c, k, rec)
                        best = min(best, (ms, (bpl, rec, k, c)))
    print("best for d=%d n=%d: %s" % (d, n, best), flush=True)
