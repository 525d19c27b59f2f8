#!/usr/bin/env python3
"""Runs bench.py with the given arguments and prints only the figures of its `summary` and of the Murray sweeps (the whole
line is a screenful).  usage: python tools/bench_brief.py --config c5 [...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
if out.returncode:
    sys.stderr.write(out.stderr[-2000:])
    sys.exit(out.returncode)
line = json.loads(out.stdout.strip().splitlines()[-1])
print("config %s: %.3f ms per job, %.4g %s" % (line["config"].get("name"), line["ms_per_step"], line["value"], line["unit"]))
mr = line.get("murray_roofline") or {}
if mr:
    print("   sweeps: %.2f ms of %.2f ms in genRemote over %s remote steps, pairs evaluated %.4f" % (
        mr.get("total_ms", 0), mr.get("whole_genremote_ms", 0), mr.get("remote_steps"), mr.get("pairs_evaluated_frac", 0)))
    sc = mr.get("screen")
    if sc:
        print("   screen: %s %.2f ms in %d launches, %.0f TFLOP/s bf16 = %.2f of the dense peak" % (
            sc["kernel"], sc["total_ms"], sc["launches"], sc["achieved"], sc["frac"]))
print("   " + json.dumps({k: v for k, v in line["summary"].items() if v is not None}))
