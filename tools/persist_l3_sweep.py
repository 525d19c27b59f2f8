#!/usr/bin/env python3
"""One-launch small-n kernel with three or more sets of chains per workgroup: equal shares per wavefront against equal
work per SIMD, by steps per phase.  One process per point."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from persist_bpl_probe import one  # noqa: E402

for d, n, bpl in ((16, 12288, 1), (16, 16384, 2), (16, 16384, 1), (32, 8192, 2), (16, 24576, 2), (16, 20480, 1)):
    for pw in (1, 0):
        for k in (12, 16, 20, 24):
            os.environ["MCX_PERSIST_COST"] = "185,110,34,24,0,0,%d" % pw
            r = one(d, n, bpl, k, 0)
            print("d=%d n=%d bpl=%d per_wave=%d K=%d -> %s" % (d, n, bpl, pw, k, r), flush=True)
