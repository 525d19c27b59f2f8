#!/usr/bin/env python3
"""One-launch small-n kernel: job time by blocks per lane, steps per phase and recorders, one process per configuration
(MCX_PERSIST_KSTEPS / MCX_PERSIST_REC are read once per process).  usage: persist_bpl_probe.py [d n]..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mcpar_amd as M
from mcpar_amd import engine as E
from persist_sweep import pinit
import time
d, n, bpl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
e = M.Engine(d, n, pl=1.0)
e.set_option(E.OPT_PERSIST, 1)
e.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
e.stage_pinit(pinit(d, n))
for _ in range(5):
    e.run(1000, 500, None, vl)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30):
        e.run(1000, 500, None, vl)
    best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
c = e.counters
print(json.dumps(dict(ms=round(best, 4), bpl=int(c["small_n_blocks_per_lane"]), launches=int(c["kernel_launches"]), to=int(c["meet_timeouts_total"]))))
''' % (ROOT, os.path.join(ROOT, "tools"))


def one(d, n, bpl, k=None, rec=None):
    env = dict(os.environ)
    if k is not None:
        env["MCX_PERSIST_KSTEPS"] = str(k)
    if rec is not None:
        env["MCX_PERSIST_REC"] = str(rec)
    o = subprocess.run([sys.executable, "-c", CHILD, str(d), str(n), str(bpl)], env=env, capture_output=True, text=True, timeout=300)
    if o.returncode != 0:
        return dict(error=o.stderr[-300:])
    return json.loads(o.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    shapes = [(16, 8192), (16, 16384), (8, 16384), (16, 12288)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for d, n in shapes:
        for bpl in (1, 2, 4):
            if d % (4 * bpl) or (bpl == 4 and d < 16):
                continue
            for rec in (1, 0):
                for k in (None, 8, 12, 14, 16, 20, 24, 32):
                    r = one(d, n, bpl, k, rec)
                    print("d=%d n=%d bpl=%d rec=%d K=%s -> %s" % (d, n, bpl, rec, k, r), flush=True)
