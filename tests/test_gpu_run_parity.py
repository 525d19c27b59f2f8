"""HIP engine (through the C ABI) vs the CPU oracle on identical seeded inputs: bit-exact.

Rows of SURVEY §8a covered: a1 burn-in loop + tuner, a2 main loop, a3 genLocal, a5 accept,
a6 sample emit, a7 Welford/adoption/publish, a4 genRemote (pl < 1), a9 covar_setup, a11 host VLFunc.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def mix_params(d, K):
    means = np.zeros((K, d), np.float32)
    for k in range(K):
        means[k, :] = 5.0 * k / max(K - 1, 1)
    w = np.ones(K, np.float32)
    w[0] = 5.0
    return np.concatenate([means.ravel(), w])


def run_pair(kind, d, n, nburn, nsamp, pl, params=None, ncomp=0, incov=None, fuse=1, sync=10,
             pinit=None, maxseg=None, host=False):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    p = O.default_pinit(d, n) if pinit is None else pinit
    vo, keep_o = O.make_vlfunc(kind, d, params, ncomp)
    eo = O.Engine(d, n, pl=pl, sync=sync, threads=4)
    eo.run(nsamp, nburn, p, vo, incov)
    if host:
        def fn(x):
            return O.vl_eval(kind, d, x, params, ncomp)
        vg, keep_g = M.make_vlfunc(M.VL_HOST, d, host_fn=fn)
    else:
        vg, keep_g = M.make_vlfunc(kind, d, params, ncomp)
    eg = M.Engine(d, n, pl=pl, sync=sync)
    eg.set_option(E.OPT_ACCEPT_MASK, 1)
    eg.set_option(E.OPT_FUSE, fuse)
    if maxseg:
        eg.set_option(E.OPT_MAX_SEGMENT, maxseg)
    eg.run(nsamp, nburn, p, vg, incov)
    return eo, eg


def assert_same(eo, eg, what=""):
    mo, mg = eo.accept_mask, eg.accept_mask
    bad = np.argwhere(mo != mg)
    assert bad.size == 0, "%s accept mask differs first at (step, chain) %s" % (what, bad[:3])
    c = eg.counters
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
    assert c["remote_steps"] == eo.remote_steps and c["remote_passes"] == eo.remote_passes
    np.testing.assert_array_equal(eg.accept_counts, eo.accept_counts)
    np.testing.assert_array_equal(eg.tuner_trace.view(np.uint32), eo.tuner_trace.view(np.uint32))
    for name in ("state", "loglike", "mean", "var", "chol", "musigall", "samples"):
        a, b = getattr(eg, name), getattr(eo, name)
        if name == "samples":  # the oracle appends across run() calls like MCout::newsamps; the
            b = b[b.shape[0] - a.shape[0]:]  # engine's HBM store holds the last run
        assert a.shape == b.shape, name
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
            "%s %s differs: max abs %g" % (what, name, np.nanmax(np.abs(a - b)))


CASES = [
    # kind, d, n, nburn, nsamp, pl
    ("dgauss_c1", O.VL_DUALGAUSS, 2, 4, 500, 8, 0.9),
    ("rosen1_d2", O.VL_ROSENBROCK1, 2, 100, 120, 30, 1.0),
    ("rosen1_d4", O.VL_ROSENBROCK1, 4, 333, 120, 30, 1.0),
    ("rosen1_d8", O.VL_ROSENBROCK1, 8, 4096, 200, 50, 1.0),
    ("rosen1_d16", O.VL_ROSENBROCK1, 16, 1000, 160, 60, 1.0),
    ("rosen1_d32", O.VL_ROSENBROCK1, 32, 257, 110, 25, 1.0),
    ("rosen1_d6_ragged", O.VL_ROSENBROCK1, 6, 77, 110, 25, 1.0),
    ("rosen1_d12_ragged", O.VL_ROSENBROCK1, 12, 65, 110, 25, 1.0),
    ("rosen1_d30_ragged", O.VL_ROSENBROCK1, 30, 19, 60, 25, 1.0),
    ("gauss_d3", O.VL_GAUSSIAN, 3, 50, 120, 40, 1.0),
    ("gauss_d2_murray", O.VL_GAUSSIAN, 2, 64, 120, 80, 0.8),
    ("rosen1_d16_murray", O.VL_ROSENBROCK1, 16, 256, 120, 60, 0.8),
    ("rosen1_d8_murray", O.VL_ROSENBROCK1, 8, 100, 120, 60, 0.7),
    # np > 32: 16 / 32 / 64 lanes per chain, Murray kernels with the chain vector in memory
    ("rosen1_d36", O.VL_ROSENBROCK1, 36, 50, 60, 20, 1.0),
    ("rosen1_d64_murray", O.VL_ROSENBROCK1, 64, 40, 110, 30, 0.8),
    ("gauss_d45_murray", O.VL_GAUSSIAN, 45, 33, 110, 30, 0.8),
    ("rosen1_d100", O.VL_ROSENBROCK1, 100, 21, 60, 12, 0.9),
    ("rosen1_d256", O.VL_ROSENBROCK1, 256, 9, 60, 12, 0.9),
]


@pytest.mark.parametrize("name,kind,d,n,nburn,nsamp,pl", CASES, ids=[c[0] for c in CASES])
def test_run_bit_exact(name, kind, d, n, nburn, nsamp, pl):
    params = None
    pinit = None
    if kind == O.VL_DUALGAUSS:
        params = [5.0]
        pinit = np.array([0, 0, 2, 2, 0, 1.5, 0, -2], np.float32)  # src/mcpar-dgauss.cc:33
    if kind == O.VL_GAUSSIAN:
        params = np.concatenate([np.linspace(-1, 1, d), np.linspace(0.5, 2, d)]).astype(np.float32)
    eo, eg = run_pair(kind, d, n, nburn, nsamp, pl, params=params, pinit=pinit)
    assert_same(eo, eg, name)


def test_mixture_c5_shape():
    d, K = 32, 8
    eo, eg = run_pair(O.VL_GAUSSMIX, d, 128, 120, 40, 0.8, params=mix_params(d, K), ncomp=K)
    assert_same(eo, eg, "mix")


def test_unfused_equals_fused():
    eo, eg = run_pair(O.VL_ROSENBROCK1, 16, 300, 120, 40, 0.9, fuse=0)
    assert_same(eo, eg, "unfused")


def test_short_segments():
    eo, eg = run_pair(O.VL_ROSENBROCK1, 8, 300, 120, 40, 1.0, maxseg=7)
    assert_same(eo, eg, "maxseg7")


def test_host_callback_vlfunc():
    eo, eg = run_pair(O.VL_ROSENBROCK1, 8, 64, 60, 20, 0.8, host=True)
    assert_same(eo, eg, "host")


def test_rosenbrock2_as_written_unfused():
    eo, eg = run_pair(O.VL_ROSENBROCK2, 4, 32, 60, 10, 1.0)
    assert_same(eo, eg, "rosen2")


def test_full_covariance():
    d = 8
    rng = np.random.default_rng(3)
    a = rng.normal(size=(d, d)).astype(np.float32)
    cov = (a @ a.T / d + 0.5 * np.eye(d)).astype(np.float32)
    eo, eg = run_pair(O.VL_ROSENBROCK1, d, 200, 120, 40, 0.9, incov=cov)
    assert_same(eo, eg, "fullcov")
    # ragged lane groups (d = 12: the fourth lane of a chain owns nothing) and 8 lanes per chain
    for d, n in ((12, 70), (32, 40), (6, 50)):
        a = rng.normal(size=(d, d)).astype(np.float32)
        cov = (a @ a.T / d + 0.5 * np.eye(d)).astype(np.float32)
        eo, eg = run_pair(O.VL_ROSENBROCK1, d, n, 60, 25, 0.9, incov=cov)
        assert_same(eo, eg, "fullcov%d" % d)
        eo, eg = run_pair(O.VL_ROSENBROCK1, d, n, 60, 25, 0.9, incov=cov, fuse=0)
        assert_same(eo, eg, "fullcov%d unfused" % d)
    # np > 32: factor read from L2 instead of LDS
    d = 40
    a = rng.normal(size=(d, d)).astype(np.float32)
    cov = (a @ a.T / d + 0.5 * np.eye(d)).astype(np.float32)
    eo, eg = run_pair(O.VL_ROSENBROCK1, d, 24, 60, 12, 0.9, incov=cov)
    assert_same(eo, eg, "fullcov40")
    # and d = 16 with 4 lanes per chain
    d = 16
    a = rng.normal(size=(d, d)).astype(np.float32)
    cov = (a @ a.T / d + 0.5 * np.eye(d)).astype(np.float32)
    eo, eg = run_pair(O.VL_ROSENBROCK1, d, 100, 60, 30, 1.0, incov=cov)
    assert_same(eo, eg, "fullcov16")


@pytest.mark.parametrize("d", [4, 8, 12, 16, 20, 24, 28, 32])
def test_full_covariance_fast_kernel(d):
    """k_fused_fast<..., FULL>: the Cholesky factor staged in LDS, z by DPP (<= 4 lanes per chain) or through LDS
    (8 lanes), every likelihood of the fast path, burn-in (tuner rescales the factor) + main loop."""
    rng = np.random.default_rng(100 + d)
    a = rng.normal(size=(d, d)).astype(np.float32)
    cov = (0.3 * (a @ a.T / d + 0.5 * np.eye(d))).astype(np.float32)
    n = 3 * 64 + 5  # several wavefronts, the last one ragged
    eo, eg = run_pair(O.VL_ROSENBROCK1, d, n, 160, 45, 1.0, incov=cov)
    assert_same(eo, eg, "fullcov fast rosen1 d%d" % d)
    g = np.concatenate([rng.normal(size=d), rng.uniform(0.5, 2.0, size=d)]).astype(np.float32)
    eo, eg = run_pair(O.VL_GAUSSIAN, d, n, 110, 30, 0.9, params=g, incov=cov)
    assert_same(eo, eg, "fullcov fast gauss d%d" % d)
    K = 3
    m = np.concatenate([rng.normal(size=K * d), [2.0, 1.0, 1.0]]).astype(np.float32)
    eo, eg = run_pair(O.VL_GAUSSMIX, d, n, 60, 30, 1.0, params=m, ncomp=K, incov=cov)
    assert_same(eo, eg, "fullcov fast mix d%d" % d)


def test_second_run_continues_rng():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n = 8, 64
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    vg, _k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eo, eg = O.Engine(d, n, pl=1.0), M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_ACCEPT_MASK, 1)
    for _ in range(2):
        eo.run(20, 60, p, vo)
        eg.run(20, 60, p, vg)
    assert_same(eo, eg, "second run")


def test_staged_pinit_equals_host_pinit():
    import mcpar_amd as M
    d, n = 16, 200
    p = O.default_pinit(d, n)
    vg, _k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    a, b = M.Engine(d, n, pl=0.9), M.Engine(d, n, pl=0.9)
    a.run(30, 60, p, vg)
    with pytest.raises(M.McxError):
        b.run(30, 60, None, vg)  # nothing staged yet
    b.stage_pinit(p)
    b.run(30, 60, None, vg)
    assert np.array_equal(a.state.view(np.uint32), b.state.view(np.uint32))
    assert np.array_equal(a.samples.view(np.uint32), b.samples.view(np.uint32))


@pytest.mark.parametrize("fuse", [1, 0])
def test_sample_stride_thins_the_store(fuse):
    """MCX_OPT_SAMPLE_STRIDE = k keeps main-loop steps 0, k, 2k, ... (same chain, same bits)"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp, k = 16, 130, 60, 47, 3
    p = O.default_pinit(d, n)
    vo, _k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=0.85)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=0.85)
    eg.set_option(E.OPT_SAMPLE_STRIDE, k)
    eg.set_option(E.OPT_FUSE, fuse)
    eg.run(nsamp, nburn, p, vg)
    want = eo.samples.reshape(nsamp, n, d + 1)[::k].reshape(-1, d + 1)
    got = eg.samples
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(eg.state.view(np.uint32), eo.state.view(np.uint32))
    # partial ranges of the store
    part = eg.samples_range(2, 5)
    assert np.array_equal(part, want[2 * n:7 * n])


@pytest.mark.parametrize("kind,d", [(O.VL_ROSENBROCK1, 16), (O.VL_GAUSSIAN, 8), (O.VL_GAUSSIAN, 32), (O.VL_GAUSSIAN, 12),
                                    (O.VL_GAUSSMIX, 32), (O.VL_GAUSSMIX, 8), (O.VL_GAUSSMIX, 4)])
def test_fast_kernel_without_mask_recording(kind, d):
    """the hot-path kernel is used only when no accept mask is recorded: compare everything else"""
    import mcpar_amd as M
    n, nburn, nsamp = 333, 120, 40
    p = O.default_pinit(d, n)
    params = None
    if kind == O.VL_GAUSSIAN:
        params = np.concatenate([np.linspace(-1, 1, d), np.linspace(0.5, 2, d)]).astype(np.float32)
    K = 0
    if kind == O.VL_GAUSSMIX:
        K = {32: 8, 8: 3, 4: 7}[d]
        params = mix_params(d, K)
    vo, _k1 = O.make_vlfunc(kind, d, params, K)
    eo = O.Engine(d, n, pl=0.9)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(kind, d, params, K)
    eg = M.Engine(d, n, pl=0.9)
    eg.run(nsamp, nburn, p, vg)
    c = eg.counters
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
    np.testing.assert_array_equal(eg.accept_counts, eo.accept_counts)
    for name in ("state", "loglike", "mean", "var", "musigall", "samples", "chol"):
        assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name


@pytest.mark.parametrize("d,n,nburn,nsamp,pl", [(1, 1, 60, 20, 0.8), (1, 37, 0, 25, 0.8), (2, 1, 120, 0, 1.0),
                                                (3, 5, 0, 0, 0.9), (4, 64, 1, 1, 1.0), (16, 3, 53, 11, 0.5)])
def test_degenerate_sizes(d, n, nburn, nsamp, pl):
    """one chain, one parameter, no burn-in, no samples: the loops of src/mcpar.cc:55-210 with empty ranges"""
    kind = O.VL_GAUSSIAN if d % 2 else O.VL_ROSENBROCK1
    eo, eg = run_pair(kind, d, n, nburn, nsamp, pl)
    assert_same(eo, eg, "degenerate")


@pytest.mark.parametrize("split,persist", [(0, 0), (1, 0), (1, 1)], ids=["fused", "pregen", "persistent"])
@pytest.mark.parametrize("kind,d,n,nburn,nsamp,pl", [(O.VL_ROSENBROCK1, 16, 333, 120, 150, 0.9), (O.VL_ROSENBROCK1, 8, 70, 59, 66, 1.0),
                                                     (O.VL_GAUSSIAN, 12, 40, 110, 129, 0.8), (O.VL_GAUSSMIX, 32, 64, 60, 70, 0.9),
                                                     (O.VL_ROSENBROCK1, 4, 1, 130, 3, 1.0), (O.VL_ROSENBROCK1, 16, 5000, 230, 77, 1.0),
                                                     (O.VL_ROSENBROCK1, 8, 9000, 52, 40, 0.9)])
def test_small_n_mode_split_rng(split, persist, kind, d, n, nburn, nsamp, pl):
    """The three forms of a stretch of local steps: random numbers generated in the step kernel (fused), by a
    separate kernel and streamed in (MCX_OPT_SPLIT_RNG), or by the generator wavefronts of the one-launch
    kernel k_run_small into LDS (MCX_OPT_PERSIST, 1-3 owner wavefronts per workgroup here).  Segment lengths
    straddle the chunks, the LDS phases and the 4-step accept blocks"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    p = O.default_pinit(d, n)
    params, K = None, 0
    if kind == O.VL_GAUSSIAN:
        params = np.concatenate([np.linspace(-1, 1, d), np.linspace(0.5, 2, d)]).astype(np.float32)
    if kind == O.VL_GAUSSMIX:
        K = 8
        params = mix_params(d, K)
    vo, _k1 = O.make_vlfunc(kind, d, params, K)
    eo = O.Engine(d, n, pl=pl)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(kind, d, params, K)
    eg = M.Engine(d, n, pl=pl)
    eg.set_option(E.OPT_SPLIT_RNG, split)
    eg.set_option(E.OPT_PERSIST, persist)
    eg.set_option(E.OPT_SAMPLE_STRIDE, 1 + split)  # also the thinned store through the chunked launches
    eg.run(nsamp, nburn, p, vg)
    c = eg.counters
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
    np.testing.assert_array_equal(eg.accept_counts, eo.accept_counts)
    assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
    for name in ("state", "loglike", "mean", "var", "musigall", "chol"):
        assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name
    want = eo.samples.reshape(nsamp, n, d + 1)[::1 + split].reshape(-1, d + 1)
    assert np.array_equal(eg.samples.view(np.uint32), want.view(np.uint32))


def test_np_bound_is_256_and_says_so():
    """The one size limit the reference does not have (src/mcpar.hh:32 takes any np): a chain's parameters live in the lanes
    of ONE wavefront, 4 per lane, so np <= 256 (include/mcx.h MCX_ERR_UNSUPPORTED; INTEGRATION.md "drop-in caveats").
    np = 256 itself is a whole, bit-exact run; 257 is refused at construction with a message -- never a wrong answer."""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 256, 48, 60, 25
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=0.85, threads=4)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=0.85)
    eg.set_option(E.OPT_ACCEPT_MASK, 1)
    eg.run(nsamp, nburn, p, vg)
    assert_same(eo, eg, "np = 256")
    eg.close()
    with pytest.raises(M.McxError) as ei:
        M.Engine(257, 8)
    assert ei.value.code == 4 and "257" in str(ei.value)  # MCX_ERR_UNSUPPORTED
