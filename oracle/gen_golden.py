#!/usr/bin/env python3
"""Generate tests/golden/*.json (run in the build container, where /root/reference exists).

  vlfunc_reference.json  inputs + outputs of the REFERENCE's own likelihood functors
                         (src/rosenbrock.cc compiled in place into oracle/_ref/ by oracle/Makefile,
                         reference Makefile flags -O3 -ffast-math).  Pins oracle + HIP likelihoods.
  mcout_reference.json   scripts + transcripts of the REFERENCE's own MCout (src/mcout.cc compiled in place
                         against the image's real MPI headers into oracle/_ref/ref_mcout_script, driven by
                         tests/cpp/mcout_script.cc) on 1 rank and under `mpiexec -n 2`: text format, incremental
                         dumps, collect() order, rewind, maxlike.  Pins the facade's MCout.
  oracle_runs.json       small end-to-end runs of the CPU oracle (oracle/mcx_oracle.c), pinning
                         the MCX arithmetic so that neither oracle nor kernels can drift silently.

Only data is written: inputs and expected outputs.
"""
import ctypes as C
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

fp = C.POINTER(C.c_float)


def f32list(a):
    return [float(v) for v in np.asarray(a, np.float32).ravel()]


def gen_vlfunc():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_vlfunc.so"))
    R.ref_dualgaussian.argtypes = [C.c_float, C.c_int, fp, fp]
    rng = np.random.default_rng(20261003)
    cases = []

    def add(name, d, x, y, **kw):
        cases.append(dict(name=name, d=d, npset=int(x.shape[0]), x=f32list(x), y=f32list(y), **kw))

    for d in (2, 4, 8, 16, 32):
        x = rng.normal(0.0, 1.2, (24, d)).astype(np.float32)
        x[0] = 1.0  # known answer 0
        x[1] = 0.0  # known answer -d/2
        y = np.empty(24, np.float32)
        assert R.ref_rosenbrock1(d, 24, O.fptr(x), O.fptr(y)) == 0
        add("rosenbrock1", d, x, y)
        y = np.empty(24, np.float32)
        assert R.ref_rosenbrock2(d, 24, O.fptr(x), O.fptr(y)) == 0
        add("rosenbrock2", d, x, y)
    # ragged / edge: one set, odd d for rosenbrock2, empty batch
    x = rng.normal(0, 1, (1, 3)).astype(np.float32)
    y = np.empty(1, np.float32)
    R.ref_rosenbrock2(3, 1, O.fptr(x), O.fptr(y))
    add("rosenbrock2", 3, x, y)
    mu = np.array([0.25, -1.5], np.float32)
    s2 = np.array([1.0, 2.0], np.float32)
    x = rng.normal(0, 3, (32, 2)).astype(np.float32)
    x[0] = mu
    y = np.empty(32, np.float32)
    assert R.ref_gaussian(2, O.fptr(mu), O.fptr(s2), 32, O.fptr(x), O.fptr(y)) == 0
    add("gaussian", 2, x, y, mu=f32list(mu), sig2=f32list(s2))
    x = rng.uniform(-4, 9, (48, 2)).astype(np.float32)
    x[0] = (0, 0)
    x[1] = (5, 5)
    y = np.empty(48, np.float32)
    R.ref_dualgaussian(5.0, 48, O.fptr(x), O.fptr(y))
    add("dualgaussian", 2, x, y, w=5.0)
    # constructor guards (src/rosenbrock.hh:14,28,43): status -1 == reference threw
    guards = dict(rosenbrock1_d3=int(R.ref_rosenbrock1(3, 0, None, None)),
                  rosenbrock1_d0=int(R.ref_rosenbrock1(0, 0, None, None)),
                  rosenbrock2_d1=int(R.ref_rosenbrock2(1, 0, None, None)),
                  gaussian_d3=int(R.ref_gaussian(3, None, None, 0, None, None)))
    return dict(source="oracle/_ref/libref_vlfunc.so built from /root/reference/src/rosenbrock.cc "
                       "(g++ -O3 -ffast-math -ftree-vectorize -Drestrict=__restrict__)",
                cases=cases, ctor_guards=guards)


MPIEXEC = "/opt/conda/bin/mpiexec"


def hexf(v):
    """float32 bit pattern as the script language writes it; strings pass through (already hex)"""
    if isinstance(v, str):
        return v
    return "%08x" % struct.unpack("<I", struct.pack("<f", v))[0]


NAN, NINF, PINF, NZERO, DENORM = "7fc00000", "ff800000", "7f800000", "80000000", "00011c37"


def mcout_scenarios():
    """each: name, np, one script (list of lines) per rank.  Collective calls (output, collect, maxlike) appear
    at the same position in every rank's script, like in MCPar::run."""
    def add(*v):
        return "add " + " ".join(hexf(x) for x in v)
    sc = []
    # 1 rank: the formats ostream << float produces, incremental dumps, rewind, equal maxima, NaN / -inf
    sc.append(dict(name="one_rank_formats", np=2, scripts=[[
        "stat", "output", "collect", "new 3", add(1.5, -2.0, -3.25), add(0.1, 1e-5, 0.5), "stat", "output", "output",
        add(123456.0, 1e10, 0.25), "output", "stat", "new 7", add(1234567.0, -1e-5, 0.5), add(NAN, NINF, NAN),
        add(PINF, NZERO, NINF), add(DENORM, 3.4e38, -1e-30), add(0.333333343, 100000.0, 0.4999999),
        add(1e6, 999999.0, 0.5), add(-0.0009765625, 16777216.0, 0.2), "output", "row 1", "row 4", "row 5", "maxlike",
        "rewind", "collect", "collect", "rewind", "output", "stat"]]))
    sc.append(dict(name="one_rank_np1_and_np5", np=1, scripts=[[
        "new 4", add(2.5, -1.0), add(-7.0, -0.5), "output", add(0.0, -0.5), add(1.0, NINF), "maxlike", "output", "stat"]]))
    sc.append(dict(name="one_rank_np5", np=5, scripts=[[
        "new 2", add(1, 2, 3, 4, 5, -6), add(0.5, 0.25, 0.125, 0.0625, 0.03125, -0.015625), "output", "maxlike",
        "rewind", "collect"]]))
    # 2 ranks: rank-major row order within a dump, several dumps, MAXLOC ties go to the lower rank
    r0 = ["new 4", add(1.0, 2.0, -1.0), add(3.0, 4.0, -0.5), "output", add(5.0, 6.0, 0.75), add(7.0, 8.0, -2.0),
          "stat", "output", "maxlike", "rewind", "collect", "output", "collect"]
    r1 = ["new 4", add(-1.0, -2.0, -3.0), add(-3.0, -4.0, 0.75), "output", add(-5.0, -6.0, 0.5), add(-7.0, -8.0, NAN),
          "stat", "output", "maxlike", "rewind", "collect", "output", "collect"]
    sc.append(dict(name="two_ranks_tie_goes_to_rank0", np=2, scripts=[r0, r1]))
    r0 = ["new 2", add(0.1, 0.2, 0.3, -10.0), "output", add(1e-5, 123456.0, 1e10, -9.0), "output", "maxlike"]
    r1 = ["new 2", add(0.4, 0.5, 0.6, -8.5), "output", add(1234567.0, NAN, NINF, -9.5), "output", "maxlike"]
    sc.append(dict(name="two_ranks_rank1_wins", np=3, scripts=[r0, r1]))
    return sc


def gen_mcout():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_mcout_script")
    out = []
    for sc in mcout_scenarios():
        nr = len(sc["scripts"])
        with tempfile.TemporaryDirectory() as td:
            files = []
            for r, lines in enumerate(sc["scripts"]):
                fn = os.path.join(td, "script.%d" % r)
                with open(fn, "w") as f:
                    f.write("\n".join(lines) + "\n")
                files.append(fn)
            res = subprocess.run([MPIEXEC, "-n", str(nr), exe, str(sc["np"])] + files, capture_output=True,
                                 text=True, timeout=120)
            assert res.returncode == 0, res.stderr
        out.append(dict(name=sc["name"], np=sc["np"], nranks=nr, scripts=sc["scripts"], transcript=res.stdout))
    return dict(source="oracle/_ref/ref_mcout_script = tests/cpp/mcout_script.cc + /root/reference/src/mcout.cc "
                       "(g++ -O2, MPICH headers of the image), run under mpiexec -n <nranks>",
                scenarios=out)


def gen_oracle_runs():
    runs = []

    def one(name, kind, d, n, nburn, nsamp, pl, params=None, ncomp=0, pinit=None, nshards=1, sync=10):
        vl, keep = O.make_vlfunc(kind, d, params, ncomp)
        engs = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl, sync=sync) for s in range(nshards)]
        pin = [O.default_pinit(d, n, g0=s * n) if pinit is None else pinit for s in range(nshards)]
        O.run_all(engs, nsamp, nburn, pin, vl)
        for s, e in enumerate(engs):
            rec = dict(name=name, kind=kind, d=d, n=n, nburn=nburn, nsamp=nsamp, pl=pl, sync=sync,
                       nshards=nshards, shard=s, params=None if params is None else f32list(params),
                       ncomp=ncomp, pinit=f32list(pin[s]),
                       naccept_burn=e.naccept_burn, naccept_main=e.naccept_main,
                       remote_steps=e.remote_steps, remote_passes=e.remote_passes,
                       tuner_trace=f32list(e.tuner_trace),
                       accept_counts=[int(v) for v in e.accept_counts],
                       state=f32list(e.state), loglike=f32list(e.loglike),
                       mean=f32list(e.mean), var=f32list(e.var))
            if n * nsamp <= 64:
                rec["samples"] = f32list(e.samples)
            runs.append(rec)

    one("dgauss_c1", O.VL_DUALGAUSS, 2, 4, 500, 8, 0.9, params=[5.0],
        pinit=np.array([0, 0, 2, 2, 0, 1.5, 0, -2], np.float32))  # src/mcpar-dgauss.cc:31-35
    one("rosen1_d8_local", O.VL_ROSENBROCK1, 8, 64, 120, 40, 1.0)
    one("rosen1_d16_murray", O.VL_ROSENBROCK1, 16, 64, 120, 60, 0.7)
    one("rosen1_d16_murray_2shards", O.VL_ROSENBROCK1, 16, 32, 120, 60, 0.7, nshards=2)
    means = np.zeros((8, 32), np.float32)
    for k in range(8):
        means[k, :] = 5.0 * k / 7.0
    w = np.array([5, 1, 1, 1, 1, 1, 1, 1], np.float32)
    one("mix_c5_small", O.VL_GAUSSMIX, 32, 32, 60, 30, 0.8, params=np.concatenate([means.ravel(), w]),
        ncomp=8)
    one("rosen2fixed_d16_local", O.VL_ROSENBROCK2_FIXED, 16, 64, 120, 40, 1.0)
    # the flagged well-posed overlapping Rosenbrock (not in the reference): its values as the oracle computes them
    rng = np.random.default_rng(20261004)
    vl_cases = []
    for d in (2, 3, 7, 16, 18):
        x = rng.normal(0.0, 1.1, (12, d)).astype(np.float32)
        x[0] = 1.0
        x[1] = 0.0
        vl_cases.append(dict(name="rosenbrock2_fixed", d=d, npset=12, x=f32list(x),
                             y=f32list(O.vl_eval(O.VL_ROSENBROCK2_FIXED, d, x))))
    return dict(source="oracle/mcx_oracle.c (MCX arithmetic v3), seed 8675309", runs=runs, vlfunc_cases=vl_cases)


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "vlfunc_reference.json"), "w") as f:
        json.dump(gen_vlfunc(), f)
    with open(os.path.join(out, "mcout_reference.json"), "w") as f:
        json.dump(gen_mcout(), f, indent=1)
    with open(os.path.join(out, "oracle_runs.json"), "w") as f:
        json.dump(gen_oracle_runs(), f)
    print("wrote", os.listdir(out))
