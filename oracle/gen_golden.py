#!/usr/bin/env python3
"""Generate tests/golden/*.json (run in the build container, where /root/reference exists).

  vlfunc_reference.json  inputs + outputs of the REFERENCE's own likelihood functors
                         (src/rosenbrock.cc compiled in place into oracle/_ref/ by oracle/Makefile,
                         reference Makefile flags -O3 -ffast-math).  Pins oracle + HIP likelihoods.
  oracle_runs.json       small end-to-end runs of the CPU oracle (oracle/mcx_oracle.c), pinning
                         the MCX arithmetic so that neither oracle nor kernels can drift silently.

Only data is written: inputs and expected outputs.
"""
import ctypes as C
import json
import os
import sys

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
    return dict(source="oracle/mcx_oracle.c (MCX arithmetic v3), seed 8675309", runs=runs)


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "vlfunc_reference.json"), "w") as f:
        json.dump(gen_vlfunc(), f)
    with open(os.path.join(out, "oracle_runs.json"), "w") as f:
        json.dump(gen_oracle_runs(), f)
    print("wrote", os.listdir(out))
