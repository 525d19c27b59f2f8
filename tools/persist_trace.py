#!/usr/bin/env python3
"""Per-wavefront timeline of the one-launch small-n kernel (a -DMCX_PERSIST_TRACE build of libmcx: build/trace/libmcx.so,
made by `tools/persist_trace.py build` in the build container).  For the first workgroups and phases: how long every
wavefront worked in a phase and how long it then waited at the phase's barrier -- who is the slowest.
usage: persist_trace.py build | persist_trace.py d n bpl [K]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    out = os.path.join(ROOT, "build", "trace")
    os.makedirs(out, exist_ok=True)
    subprocess.check_call(["make", "-C", ROOT, "lib"], stdout=subprocess.DEVNULL)
    flags = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Iinclude".split()
    subprocess.check_call(["hipcc"] + flags + ["-DMCX_PERSIST_TRACE", "-c", "-o", out + "/persist.o", "mcpar_amd/csrc/mcx_k_persist.hip"], cwd=ROOT)
    objs = [os.path.join(ROOT, "mcpar_amd", "csrc", f) for f in sorted(os.listdir(os.path.join(ROOT, "mcpar_amd", "csrc")))
            if f.endswith(".o") and f != "mcx_k_persist.o"]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out + "/libmcx.so"] + objs + [out + "/persist.o"])
    os.remove(out + "/persist.o")
    print("built", out + "/libmcx.so")


def main():
    if sys.argv[1:] == ["build"]:
        return build()
    d, n, bpl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4:
        os.environ["MCX_PERSIST_KSTEPS"] = sys.argv[4]
    os.environ["MCX_LIBMCX"] = os.path.join(ROOT, "build", "trace", "libmcx.so")
    dump = "/tmp/mcx_persist_trace.bin"
    os.environ["MCX_PERSIST_TRACE"] = dump
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import mcpar_amd as M
    from mcpar_amd import engine as E
    from persist_sweep import pinit
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    e = M.Engine(d, n, pl=1.0)
    e.set_option(E.OPT_PERSIST, 1)
    e.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
    e.stage_pinit(pinit(d, n))
    for _ in range(4):
        e.run(1000, 500, None, vl)
    t = np.fromfile(dump, dtype=np.uint64).reshape(4, 16, 128, 2).astype(np.int64)
    nph = int((t[0, 0, :, 1] > 0).sum())
    print("d=%d n=%d bpl=%d: %d phases traced, blocks per lane %d" % (d, n, bpl, nph, e.counters["small_n_blocks_per_lane"]))
    ms = t[1, 15, 126:128, :].reshape(4)  # the launch's milestones (wavefront 15's last two slots)
    if ms[0] > 0:
        print("workgroup 1: entry -> state ready %d clocks, first fill %d, phase loop %d (%.1f per phase)"
              % (ms[1] - ms[0], ms[2] - ms[1], ms[3] - ms[2], (ms[3] - ms[2]) / max(nph, 1)))
    wg = 1
    end = t[wg, :, :nph, 1]          # after the barrier (same for all waves up to skew)
    work_end = t[wg, :, :nph, 0]
    start = np.concatenate([np.full((16, 1), 0), end[:, :-1]], axis=1)
    work = (work_end - start)[:, 1:]  # phase 0 has no defined start
    wait = (end - work_end)[:, 1:]
    dur = (end[0, 1:] - end[0, :-1])
    for name, lo, hi in (("burn-in phases", 2, min(nph, 20)), ("main-loop phases", max(nph - 30, 2), nph - 2)):
        if hi <= lo:
            continue
        print("%s %d..%d: phase length %.0f clocks" % (name, lo, hi, dur[lo - 1:hi - 1].mean()))
        for w in range(16):
            print("   wave %2d (SIMD %d): works %6.0f  waits %6.0f" % (w, w % 4, work[w, lo - 1:hi - 1].mean(), wait[w, lo - 1:hi - 1].mean()))


if __name__ == "__main__":
    main()
