"""The oracle's arithmetic primitives: published known answers and accuracy against float64."""
import ctypes as C

import numpy as np

import oracle_lib as O


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_log_exp_sincos_accuracy():
    L = O.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(4000), 2.0 ** -np.arange(1, 34)]).astype(np.float32)
    for x in xs:
        ref = np.log(np.float64(x))
        assert abs(L.mcxo_logf(float(x)) - ref) <= 2e-7 * max(1.0, abs(ref))
    assert L.mcxo_logf(1.0) == 0.0
    for x in rng.uniform(-87, 88, 4000).astype(np.float32):
        ref = np.exp(np.float64(x))
        assert abs(L.mcxo_expf(float(x)) - ref) <= 3e-7 * ref
    assert L.mcxo_expf(0.0) == 1.0
    assert L.mcxo_expf(-100.0) == 0.0 and L.mcxo_expf(89.0) == float("inf")
    # results are normal floats or exactly 0: n = floor(x log2(e) + 1/2) < -125 flushes
    tiny = [L.mcxo_expf(float(x)) for x in np.linspace(-90, -86, 4001).astype(np.float32)]
    assert all(v == 0.0 or v >= 2.0 ** -126 for v in tiny) and tiny[0] == 0.0 and tiny[-1] > 0.0
    assert L.mcxo_accept_lu(0) == float("-inf") and L.mcxo_accept_lu(255) == float("-inf")
    assert abs(L.mcxo_accept_lu(0x80000000) - np.log(0.5)) < 1e-7 and L.mcxo_accept_lu(0xffffffff) < 0.0
    assert np.isnan(L.mcxo_expf(float("nan")))
    s, c = C.c_float(), C.c_float()
    for w in rng.integers(0, 2 ** 32, 4000, dtype=np.uint64):
        L.mcxo_sincos2pi(int(w), C.byref(s), C.byref(c))
        a = 2 * np.pi * int(w) / 2 ** 32
        assert abs(s.value - np.sin(a)) < 3e-7 and abs(c.value - np.cos(a)) < 3e-7


def test_exp_is_monotone_on_a_dense_sample():
    L = O.lib()
    xs = np.sort(np.random.default_rng(1).uniform(-87.3, 5, 20000).astype(np.float32))
    ys = np.array([L.mcxo_expf(float(x)) for x in xs])
    assert np.all(np.diff(ys) >= 0)


def test_uniform_ranges():
    L = O.lib()
    assert L.mcxo_u24(0) == 0.0 and L.mcxo_u24(0xffffffff) < 1.0
    assert L.mcxo_uopen(0) > 0.0 and L.mcxo_uopen(0xffffffff) == 1.0


def test_normals_are_standard():
    L = O.lib()
    z = np.zeros(4, np.float32)
    out = []
    for g in range(20000):
        L.mcxo_normal4(8675309, 0, 3, g, 1, 0, O.fptr(z))
        out.append(z.copy())
    out = np.asarray(out, np.float64)
    assert np.all(np.abs(out.mean(0)) < 0.03)
    assert np.all(np.abs(out.var(0) - 1) < 0.04)
    assert np.all(np.abs(np.corrcoef(out.T) - np.eye(4)) < 0.03)
    flat = out.ravel()
    assert abs(((flat - flat.mean()) ** 4).mean() / flat.var() ** 2 - 3.0) < 0.1


def test_cholesky_identity_and_spd():
    L = O.lib()
    a = np.eye(5, dtype=np.float32)
    assert L.mcxo_cholesky(5, O.fptr(a)) == 0
    assert np.array_equal(a, np.eye(5, dtype=np.float32))  # default covar_setup -> identity
    rng = np.random.default_rng(2)
    m = rng.normal(size=(6, 6))
    spd = (m @ m.T + 6 * np.eye(6)).astype(np.float32)
    f = spd.copy()
    assert L.mcxo_cholesky(6, O.fptr(f)) == 0
    np.testing.assert_allclose(f @ f.T, spd, rtol=2e-5, atol=2e-5)
    assert np.all(np.triu(f, 1) == 0)
    bad = -np.eye(3, dtype=np.float32)
    assert L.mcxo_cholesky(3, O.fptr(bad)) != 0


def test_vectorised_murray_sweep_equals_the_scalar_statement():
    """The oracle's Murray sweep has a plain scalar statement and an eight-at-a-time AVX2 form of the same
    IEEE operations (oracle/mcx_oracle.c: sweep_scalar / sweep_avx2).  They must agree bit for bit,
    in the many-pass regime too (d not a multiple of 4, N not a multiple of 8 or of QBLOCK)."""
    import numpy as np
    rng = np.random.default_rng(7)
    try:
        for d, n, nsh in ((16, 700, 1), (6, 301, 2), (32, 260, 1), (2, 9, 1)):
            N = n * nsh
            ms = (rng.random((N, d, 2)) + 0.5).astype(np.float32)
            pv = rng.standard_normal((n, d)).astype(np.float32)
            out = []
            for scalar in (1, 0):
                O.lib().mcxo_set_scalar_sweep(scalar)
                e = O.Engine(d, n, nshards=nsh, shard=nsh - 1, threads=4)
                out.append(e.gen_remote(11, pv, ms))
                e.close()
            assert out[0][4] == out[1][4] and out[0][4] > 3  # pass counts
            for a, b in zip(out[0][:4], out[1][:4]):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        O.lib().mcxo_set_scalar_sweep(0)


def _murray_f64(pvals, ptrial, ms):
    """float64 evaluation of the reference's own formula, xm = mu - x; arg += xm*xm/sig2 (src/mcpar.cc:369-386,
    425-436): max_i Q_i(pvals), max_i Q_i(ptrial)"""
    mu, s2 = ms[:, :, 0].astype(np.float64), ms[:, :, 1].astype(np.float64)

    def qmax(x):
        arg = (((mu[None] - x[:, None].astype(np.float64)) ** 2) / s2[None]).sum(2)
        return np.exp(-0.5 * arg.min(1))
    return qmax(pvals), qmax(ptrial)


def test_murray_sweep_keeps_the_cancellation_of_the_reference_formula():
    """(mu - x)^2 / sig2 must be formed from the exact difference.  Chains that have not moved since
    the main loop began sit exactly on their own Gaussian's mean with sig2 = FPEPS / pwgt ~ 1e-15..1e-16: the
    reference gives arg = 0, Q = 1 there (so cfac's numerator is exactly 1); and targets far from the origin
    with narrow per-chain Gaussians (|mu| / sigma ~ 1e5) must not lose the digits of mu - x either."""
    rng = np.random.default_rng(11)
    d, n = 16, 192
    # (a) never-accepted chains among ordinary ones
    ms = np.empty((n, d, 2), np.float32)
    ms[:, :, 0] = rng.normal(0.3, 0.5, (n, d))
    ms[:, :, 1] = rng.uniform(0.02, 0.2, (n, d)) ** 2
    stuck = np.arange(0, n, 3)
    ms[stuck, :, 1] = np.float32(1e-14) / np.float32(rng.integers(10, 90, (len(stuck), 1)))
    pv = (ms[:, :, 0] + np.sqrt(ms[:, :, 1]) * rng.standard_normal((n, d))).astype(np.float32)
    pv[stuck] = ms[stuck, :, 0]
    e = O.Engine(d, n, threads=4)
    pt, cf, mt, sg, npass = e.gen_remote(17, pv, ms)
    e.close()
    cmax64, qmax64 = _murray_f64(pv, pt, ms)
    assert np.all(cmax64[stuck] == 1.0)
    np.testing.assert_allclose(cf, np.maximum(cmax64, 0) / np.maximum(qmax64, 1e-14), rtol=2e-4)
    assert np.all(np.abs(cf[stuck] * qmax64[stuck] - 1.0) < 2e-4)  # numerator exactly 1 for the stuck chains
    # (b) a shifted, narrow target: |mu| ~ 1000, sigma ~ 0.01
    ms[:, :, 0] = 1000.0 + rng.normal(0, 0.05, (n, d))
    ms[:, :, 1] = rng.uniform(0.008, 0.02, (n, d)) ** 2
    pv = (ms[:, :, 0] + np.sqrt(ms[:, :, 1]) * rng.standard_normal((n, d))).astype(np.float32)
    e = O.Engine(d, n, threads=4)
    pt, cf, mt, sg, npass = e.gen_remote(23, pv, ms)
    e.close()
    cmax64, qmax64 = _murray_f64(pv, pt, ms)
    np.testing.assert_allclose(cf, cmax64 / np.maximum(qmax64, 1e-14), rtol=5e-4)
