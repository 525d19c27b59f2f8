"""Batched VLFunc evaluation on the GPU (through mcx_vlfunc_eval): bit-exact vs the oracle,
within fp32 tolerance of the reference's own functors (golden vectors), plus edge cases."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

KIND = {"rosenbrock1": 1, "rosenbrock2": 2, "gaussian": 3, "dualgaussian": 4}


def test_reference_golden_vectors(golden_dir):
    import mcpar_amd as M
    g = json.load(open(os.path.join(golden_dir, "vlfunc_reference.json")))
    for c in g["cases"]:
        params = None
        if c["name"] == "gaussian":
            params = np.array(c["mu"] + c["sig2"], np.float32)
        if c["name"] == "dualgaussian":
            params = [c["w"]]
        x = np.array(c["x"], np.float32).reshape(c["npset"], c["d"])
        y = M.vlfunc_eval(KIND[c["name"]], c["d"], x, params)
        np.testing.assert_allclose(y, np.array(c["y"], np.float32), rtol=3e-6, atol=2e-5)
        yo = O.vl_eval(KIND[c["name"]], c["d"], x, params)
        assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), c["name"]


@pytest.mark.parametrize("d", [2, 4, 6, 8, 10, 16, 22, 32, 34, 64, 130, 256])
def test_rosenbrock1_bit_exact(d):
    import mcpar_amd as M
    x = np.random.default_rng(d).normal(0, 1.5, (70001 if d <= 32 else 5001, d)).astype(np.float32)
    assert np.array_equal(M.vlfunc_eval(M.VL_ROSENBROCK1, d, x).view(np.uint32),
                          O.vl_eval(O.VL_ROSENBROCK1, d, x).view(np.uint32))


@pytest.mark.parametrize("d", [2, 3, 5, 16, 31, 77, 255])
def test_rosenbrock2_and_gaussian_bit_exact(d):
    import mcpar_amd as M
    rng = np.random.default_rng(100 + d)
    x = rng.normal(0, 1.5, (5003, d)).astype(np.float32)
    assert np.array_equal(M.vlfunc_eval(M.VL_ROSENBROCK2, d, x).view(np.uint32),
                          O.vl_eval(O.VL_ROSENBROCK2, d, x).view(np.uint32))
    params = np.concatenate([rng.normal(size=d), rng.uniform(0.3, 3, d)]).astype(np.float32)
    assert np.array_equal(M.vlfunc_eval(M.VL_GAUSSIAN, d, x, params).view(np.uint32),
                          O.vl_eval(O.VL_GAUSSIAN, d, x, params).view(np.uint32))
    assert np.array_equal(M.vlfunc_eval(M.VL_GAUSSIAN, d, x).view(np.uint32),
                          O.vl_eval(O.VL_GAUSSIAN, d, x).view(np.uint32))


def test_mixture_and_dualgaussian_bit_exact():
    import mcpar_amd as M
    rng = np.random.default_rng(9)
    x = rng.uniform(-4, 9, (4096, 2)).astype(np.float32)
    assert np.array_equal(M.vlfunc_eval(M.VL_DUALGAUSS, 2, x, [5.0]).view(np.uint32),
                          O.vl_eval(O.VL_DUALGAUSS, 2, x, [5.0]).view(np.uint32))
    d, K = 32, 8
    means = np.stack([np.full(d, 5.0 * k / 7.0) for k in range(K)]).astype(np.float32)
    params = np.concatenate([means.ravel(), [5, 1, 1, 1, 1, 1, 1, 1]]).astype(np.float32)
    x = rng.uniform(-2, 7, (3000, d)).astype(np.float32)
    assert np.array_equal(M.vlfunc_eval(M.VL_GAUSSMIX, d, x, params, K).view(np.uint32),
                          O.vl_eval(O.VL_GAUSSMIX, d, x, params, K).view(np.uint32))
    # far tails: log-sum-exp stays finite where exp() underflows
    far = np.full((4, d), 40.0, np.float32)
    assert np.all(np.isfinite(M.vlfunc_eval(M.VL_GAUSSMIX, d, far, params, K)))


def test_edge_cases_and_errors():
    import mcpar_amd as M
    assert M.vlfunc_eval(M.VL_ROSENBROCK1, 4, np.zeros((0, 4), np.float32)).shape == (0,)
    assert M.vlfunc_eval(M.VL_ROSENBROCK1, 2, [[1.0, 1.0]])[0] == 0.0
    assert M.vlfunc_eval(M.VL_ROSENBROCK1, 16, np.zeros((1, 16), np.float32))[0] == -8.0
    for kind, d in ((M.VL_ROSENBROCK1, 3), (M.VL_ROSENBROCK2, 1), (M.VL_DUALGAUSS, 3)):
        with pytest.raises(M.McxError) as ei:
            M.vlfunc_eval(kind, d, np.zeros((1, d), np.float32), [5.0])
        assert ei.value.code == 1
    x = np.array([[np.nan, 1.0], [np.inf, 0.0]], np.float32)
    a, b = M.vlfunc_eval(M.VL_ROSENBROCK1, 2, x), O.vl_eval(O.VL_ROSENBROCK1, 2, x)
    assert np.array_equal(np.isnan(a), np.isnan(b))
