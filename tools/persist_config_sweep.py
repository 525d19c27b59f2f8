#!/usr/bin/env python3
"""One-launch small-n kernel: job time by blocks per lane, recorders yes/no and steps per phase over the shapes the
automatic choice has to cover (mcx_k_persist.hip: mcxk_persist_bpl / _recorders / _ksteps).  One process per point."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from persist_bpl_probe import one  # noqa: E402

if __name__ == "__main__":
    shapes = [(8, 4096), (16, 4096), (8, 8192), (16, 8192), (8, 16384), (16, 12288), (16, 16384), (16, 24576), (32, 4096), (32, 8192)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for d, n in shapes:
        res = []
        r0 = one(d, n, 0, None, None)
        print("d=%d n=%d automatic -> %s" % (d, n, r0), flush=True)
        for bpl in (1, 2):
            if d % (4 * bpl):
                continue
            for rec in (1, 0):
                for k in (8, 12, 16, 24, 32):
                    r = one(d, n, bpl, k, rec)
                    res.append((r.get("ms", 9e9), bpl, rec, k))
                    print("d=%d n=%d bpl=%d rec=%d K=%d -> %s" % (d, n, bpl, rec, k, r), flush=True)
        print("best for d=%d n=%d: %s   (automatic: %s)" % (d, n, sorted(res)[:3], r0), flush=True)
