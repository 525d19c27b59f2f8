#!/usr/bin/env python3
"""The generator deal's cost model after the generators got cheaper (Philox on v_bitop3, select-free log): job time against
cn (a step of normals), ca (a Philox block of acceptance logs), co / cr (an owner's / a recorder's step), everything else as
the engine picks it.  usage: persist_cost_sweep2.py d n"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from persist_bpl_probe import one  # noqa: E402

d, n = int(sys.argv[1]), int(sys.argv[2])
res = []
CN = tuple(int(v) for v in os.environ.get("SWEEP_CN", "185,165,150,135,120").split(","))
COCR = tuple(tuple(int(x) for x in v.split("/")) for v in os.environ.get("SWEEP_COCR", "34/24,45/32,60/40").split(","))
for cn in CN:
    for ca in (110, 85):
        for co, cr in COCR:
            c = "%d,%d,%d,%d" % (cn, ca, co, cr)
            os.environ["MCX_PERSIST_COST"] = c
            r = one(d, n, 0)
            print("d=%d n=%d cost=%-16s -> %s" % (d, n, c, r), flush=True)
            res.append((r.get("ms", 9e9), c))
res.sort()
print("best five for d=%d n=%d: %s" % (d, n, res[:5]), flush=True)
print("default 185,110,34,24: %s" % [r for r in res if r[1] == "185,110,34,24"], flush=True)
