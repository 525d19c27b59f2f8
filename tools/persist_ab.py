#!/usr/bin/env python3
"""A/B of the one-launch small-n kernel's meeting logic on ONE box: the libraries tools/meet_ab.sh
built (build/ab/libmcx_v{0,1,2}.so) time the same jobs in turn, several rounds, each in a process of its own
(MCX_LIBMCX picks the library).  Prints the job time and the kernel's own time (HIP events, MCX_OPT_PROFILE) per variant."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, time, json
sys.path.insert(0, %r)
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E
def pinit(d, n):
    g = np.arange(n, dtype=np.float64)[:, None]; i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)
out = {}
for d, n, bpl in ((8, 4096, 0), (16, 8192, 0), (16, 4096, 0), (8, 16384, 0), (16, 16384, 0), (16, 12288, 0)):
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_PERSIST, 1)
    e.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
    e.stage_pinit(pinit(d, n))
    for _ in range(5):
        e.run(1000, 500, None, vl)
    reps = 60
    t0 = time.perf_counter()
    for _ in range(reps):
        e.run(1000, 500, None, vl)
    job = (time.perf_counter() - t0) / reps * 1e3
    e.set_option(E.OPT_PROFILE, 1)
    for _ in range(10):
        e.run(1000, 500, None, vl)
    pr = e.profile
    ker = pr["run_small"]["ms"] / max(pr["run_small"]["launches"], 1)
    out["%%dx%%d bpl%%d" %% (n, d, bpl)] = (round(job, 4), round(ker, 4), int(e.counters["small_n_blocks_per_lane"]))
    e.close()
print(json.dumps(out))
''' % ROOT


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    variants = sys.argv[2:] or ["r3", "v0", "v2"]  # build/ab/libmcx_<name>.so: round 3's library, unbounded meetings, what ships
    res = {}
    for r in range(rounds):
        for v in variants:
            lib = os.path.join(ROOT, "build", "ab", "libmcx_%s.so" % v)
            env = dict(os.environ, MCX_LIBMCX=lib)
            o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
            if o.returncode != 0:
                print("variant %s failed:\n%s" % (v, o.stderr[-2000:]))
                return 1
            got = json.loads(o.stdout.strip().splitlines()[-1])
            for k, val in got.items():
                res.setdefault(k, {}).setdefault(v, []).append(val)
            print("round %d variant %s: %s" % (r, v, got), flush=True)
    print("\nshape: variant -> min job ms / min kernel ms over %d rounds" % rounds)
    for k, byv in res.items():
        print(k, {v: (min(x[0] for x in xs), min(x[1] for x in xs)) for v, xs in byv.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
