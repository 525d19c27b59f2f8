"""mcutil::qriguess of the C++ facade (role of src/mcutil.cc): Sobol initial guesses in [plo, phi],
rank r taking points r*npset .. (r+1)*npset-1.  CPU only."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(exe, rank, npset, nparam):
    r = subprocess.run([exe, str(rank), str(npset), str(nparam)], capture_output=True, text=True, timeout=60)
    return r.returncode, r.stdout


def test_sobol_initial_guesses(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "mcpar_amd", "drivers"), "../libmcpar.so"],
                          stdout=subprocess.DEVNULL)
    exe = str(tmp_path / "mcutil_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "mcutil_check.cc"), "-o", exe,
                           "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcpar", "-lmcx",
                           "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd")])
    d, m = 21, 10
    n = 2 ** m - 1  # points 1 .. 2^m - 1 (the all-zero point 0 is skipped)
    rc, out = run(exe, 0, n, d)
    assert rc == 0
    x = np.array([[float(v) for v in line.split()] for line in out.strip().split("\n")])
    assert x.shape == (n, d)
    lo = -1.0 - np.arange(d)
    hi = 2.0 + 0.5 * np.arange(d)
    assert np.all(x >= lo - 1e-6) and np.all(x < hi)
    u = (x - lo) / (hi - lo)
    # every one-dimensional projection of the first 2^m points is the dyadic grid k / 2^m
    for i in range(d):
        k = np.sort(np.rint(u[:, i] * 2 ** m).astype(int))
        assert np.array_equal(k, np.arange(1, 2 ** m)), i
    # dimension 1 is van der Corput in Gray-code order: 1/2, 3/4, 1/4, 3/8, ...
    np.testing.assert_allclose(u[:4, 0], [0.5, 0.75, 0.25, 0.375], atol=1e-6)
    # low discrepancy in 2-D projections: 16 x 16 boxes hold 1023/256 ~ 4 points each
    for (a, b) in ((0, 1), (1, 2), (5, 9), (19, 20)):
        h, _, _ = np.histogram2d(u[:, a], u[:, b], bins=16, range=[[0, 1], [0, 1]])
        assert h.min() >= 2 and h.max() <= 6, (a, b, h.min(), h.max())
    # ranks continue the sequence: rank 1 with npset = 100 gives points 100 .. 199
    rc, a = run(exe, 0, 200, 4)
    rc, b = run(exe, 1, 100, 4)
    assert a.strip().split("\n")[100:] == b.strip().split("\n")
    rc, out = run(exe, 0, 4, 22)
    assert rc == 3 and "nparam <= 21" in out
