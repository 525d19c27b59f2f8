"""Pins the oracle: (1) against vectors produced by the REFERENCE's own likelihood functors
(tests/golden/vlfunc_reference.json, made by oracle/gen_golden.py from oracle/_ref), (2) against
the known answers of SURVEY §4, (3) against its own committed end-to-end runs so that the
arithmetic specification cannot drift."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

KIND = {"rosenbrock1": O.VL_ROSENBROCK1, "rosenbrock2": O.VL_ROSENBROCK2, "gaussian": O.VL_GAUSSIAN,
        "dualgaussian": O.VL_DUALGAUSS}


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def case_params(c):
    if c["name"] == "gaussian":
        return np.array(c["mu"] + c["sig2"], np.float32)
    if c["name"] == "dualgaussian":
        return [c["w"]]
    return None


def test_likelihoods_match_reference_vectors(golden_dir):
    g = load(golden_dir, "vlfunc_reference.json")
    assert len(g["cases"]) >= 13
    for c in g["cases"]:
        x = np.array(c["x"], np.float32).reshape(c["npset"], c["d"])
        y = O.vl_eval(KIND[c["name"]], c["d"], x, case_params(c))
        ref = np.array(c["y"], np.float32)
        # fp32, reference built with -ffast-math: a few ulp of the largest term
        np.testing.assert_allclose(y, ref, rtol=3e-6, atol=2e-5, err_msg="%s d=%d" % (c["name"], c["d"]))


def test_constructor_guards_match_reference(golden_dir):
    g = load(golden_dir, "vlfunc_reference.json")["ctor_guards"]
    assert g == {"rosenbrock1_d3": -1, "rosenbrock1_d0": -1, "rosenbrock2_d1": -1, "gaussian_d3": -1}
    with pytest.raises(ValueError):
        O.vl_eval(O.VL_ROSENBROCK1, 3, np.zeros((1, 3)))  # src/rosenbrock.hh:14
    with pytest.raises(ValueError):
        O.vl_eval(O.VL_ROSENBROCK2, 1, np.zeros((1, 1)))  # src/rosenbrock.hh:28


def test_known_answers():
    for d in (2, 8, 16):
        assert O.vl_eval(O.VL_ROSENBROCK1, d, np.ones((1, d)))[0] == 0.0
        assert O.vl_eval(O.VL_ROSENBROCK1, d, np.zeros((1, d)))[0] == -d / 2
    mu = np.array([0.5, -2.0], np.float32)
    assert O.vl_eval(O.VL_GAUSSIAN, 2, mu[None, :], np.concatenate([mu, [1, 2]]))[0] == 0.0
    y = O.vl_eval(O.VL_DUALGAUSS, 2, np.array([[0, 0], [5, 5]], np.float32), [5.0])
    np.testing.assert_allclose(y, [np.log(5 + np.exp(-25.0)), np.log(5 * np.exp(-25.0) + 1)], rtol=1e-6, atol=1e-7)


def test_rosenbrock2_fixed_known_answers_float64_and_pinned_values(golden_dir):
    """MCX_VL_ROSENBROCK2_FIXED is NOT reference behaviour (the reference's Rosenbrock2 has the sign error and the
    boundary crossing of src/rosenbrock.cc:32-38): the flagged well-posed variant, pinned by its closed form."""
    for d in (2, 5, 16):
        assert O.vl_eval(O.VL_ROSENBROCK2_FIXED, d, np.ones((1, d)))[0] == 0.0
        assert O.vl_eval(O.VL_ROSENBROCK2_FIXED, d, np.zeros((1, d)))[0] == -(d - 1)
    rng = np.random.default_rng(8)
    for d in (2, 3, 9, 16, 40):
        x = rng.normal(0, 1.2, (50, d)).astype(np.float32)
        x64 = x.astype(np.float64)
        ref = -(((1 - x64[:, :-1]) ** 2) + 100 * (x64[:, 1:] - x64[:, :-1] ** 2) ** 2).sum(1)
        np.testing.assert_allclose(O.vl_eval(O.VL_ROSENBROCK2_FIXED, d, x), ref, rtol=3e-6)
    # two sets side by side: nothing leaks across the set boundary (the reference's Rosenbrock2 does)
    x = rng.normal(0, 1, (2, 6)).astype(np.float32)
    both = O.vl_eval(O.VL_ROSENBROCK2_FIXED, 6, x)
    assert both[0] == O.vl_eval(O.VL_ROSENBROCK2_FIXED, 6, x[:1])[0]
    for c in load(golden_dir, "oracle_runs.json")["vlfunc_cases"]:
        xx = np.array(c["x"], np.float32).reshape(c["npset"], c["d"])
        assert np.array_equal(O.vl_eval(O.VL_ROSENBROCK2_FIXED, c["d"], xx), np.array(c["y"], np.float32))


def test_empty_batch():
    assert O.vl_eval(O.VL_ROSENBROCK1, 4, np.zeros((0, 4))).shape == (0,)


def test_mixture_reduces_to_dualgaussian():
    x = np.random.default_rng(5).uniform(-3, 8, (64, 2)).astype(np.float32)
    a = O.vl_eval(O.VL_DUALGAUSS, 2, x, [5.0])
    b = O.vl_eval(O.VL_GAUSSMIX, 2, x, np.array([0, 0, 5, 5, 5, 1], np.float32), ncomp=2)
    assert np.array_equal(a, b)


def test_oracle_runs_are_pinned(golden_dir):
    g = load(golden_dir, "oracle_runs.json")
    by_name = {}
    for r in g["runs"]:
        by_name.setdefault(r["name"], []).append(r)
    assert len(by_name) >= 5
    for name, recs in by_name.items():
        recs.sort(key=lambda r: r["shard"])
        r0 = recs[0]
        vl, keep = O.make_vlfunc(r0["kind"], r0["d"], r0["params"], r0["ncomp"])
        engs = [O.Engine(r0["d"], r0["n"], nshards=r0["nshards"], shard=s, pl=r0["pl"], sync=r0["sync"])
                for s in range(r0["nshards"])]
        O.run_all(engs, r0["nsamp"], r0["nburn"], [np.array(r["pinit"], np.float32) for r in recs], vl)
        for r, e in zip(recs, engs):
            assert e.naccept_burn == r["naccept_burn"] and e.naccept_main == r["naccept_main"], name
            assert e.remote_steps == r["remote_steps"] and e.remote_passes == r["remote_passes"], name
            assert [int(v) for v in e.accept_counts] == r["accept_counts"], name
            for key in ("state", "loglike", "mean", "var", "tuner_trace"):
                got = np.asarray(getattr(e, key), np.float32).ravel()
                assert np.array_equal(got, np.array(r[key], np.float32)), "%s %s" % (name, key)
            if "samples" in r:
                assert np.array_equal(e.samples.ravel(), np.array(r["samples"], np.float32)), name


def test_run_all_equals_single_shard_engine_when_one_shard():
    d, n = 8, 32
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    p = O.default_pinit(d, n)
    a = O.Engine(d, n, pl=0.8)
    a.run(40, 60, p, vl)
    b = O.Engine(d, n, pl=0.8)
    O.run_all([b], 40, 60, [p], vl)
    assert np.array_equal(a.state, b.state) and np.array_equal(a.accept_mask, b.accept_mask)
