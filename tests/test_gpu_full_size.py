"""BASELINE.json's full sizes (C2: 8-D x 4096, C3: 16-D x 65536 on one GPU) through size-independent
properties, plus exact equality with the oracle on the parts the oracle finishes in seconds."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_c3_full_size_properties_and_oracle_prefix():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 16, 65536, 500, 1000
    p = O.default_pinit(d, n)
    vg, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.run(nsamp, nburn, p, vg)
    c = eg.counters
    acc = eg.accept_counts.astype(np.int64)
    # checksum of checksums: per-chain accept counts add up to the device-wide ballot counters
    assert acc.sum() == c["naccept_burn"] + c["naccept_main"]
    rate = c["naccept_main"] / (n * nsamp)
    assert 0.15 < rate < 0.5
    # sample store: last row of every chain is the final state; loglike column is L(row)
    last = eg.samples_range(nsamp - 1, 1)
    assert np.array_equal(last[:, :d], eg.state) and np.array_equal(last[:, d], eg.loglike)
    rows = eg.samples_range(500, 2)
    ly = M.vlfunc_eval(M.VL_ROSENBROCK1, d, rows[:, :d])
    assert np.array_equal(ly.view(np.uint32), rows[:, d].view(np.uint32))
    # a row either repeats the previous step's row (rejection) or is new; rate consistent
    changed = np.any(rows[n:, :d] != rows[:n, :d], axis=1)
    assert abs(changed.mean() - rate) < 0.03
    # running moments equal the moments of the stored samples (Welford identity)
    sub = slice(0, 256)
    chunk = np.stack([eg.samples_range(s, 1)[sub, :d] for s in range(0, nsamp)]).astype(np.float64)
    np.testing.assert_allclose(eg.mean[sub], chunk.mean(0), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(eg.var[sub], chunk.var(0), rtol=5e-3, atol=1e-6)
    # idempotence: the same job again gives the same bits only if the RNG counter is reset -> a
    # fresh engine reproduces the run exactly
    eg2 = M.Engine(d, n, pl=1.0)
    eg2.run(nsamp, nburn, p, vg)
    assert np.array_equal(eg2.state.view(np.uint32), eg.state.view(np.uint32))
    assert np.array_equal(eg2.accept_counts, eg.accept_counts)
    # the oracle on the same 65 536 chains for a shorter job (seconds): bit-exact
    vo, keep2 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=16)
    eo.set_record(samples=False, mask=True)
    eo.run(30, 110, p, vo)
    eg3 = M.Engine(d, n, pl=1.0)
    eg3.set_option(E.OPT_ACCEPT_MASK, 1)
    eg3.set_option(E.OPT_SAMPLES, 0)
    eg3.run(30, 110, p, vg)
    assert np.array_equal(eg3.accept_mask, eo.accept_mask)
    for name in ("state", "loglike", "mean", "var"):
        assert np.array_equal(getattr(eg3, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name
    assert np.array_equal(eg3.tuner_trace, eo.tuner_trace)


def test_c2_full_job_bit_exact():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 8, 4096, 500, 1000
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=16)
    eo.run(nsamp, nburn, p, vo)
    vg, k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_ACCEPT_MASK, 1)
    eg.run(nsamp, nburn, p, vg)
    assert np.array_equal(eg.accept_mask, eo.accept_mask)
    assert np.array_equal(eg.samples.view(np.uint32), eo.samples.view(np.uint32))
    assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
    lm, pm = eg.maxlike()
    so = eo.samples
    assert lm == so[:, d].max() and np.array_equal(pm, so[np.argmax(so[:, d]), :d])


def test_murray_4096_chains_bit_exact():
    """R-murray parity case of SURVEY §8d (<= 4096 chains, pl = 0.9, nsamp = 100), pass counts reported"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 16, 4096, 500, 100
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=0.9, threads=16)
    eo.set_record(samples=False, mask=True)
    eo.run(nsamp, nburn, p, vo)
    vg, k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=0.9)
    eg.set_option(E.OPT_ACCEPT_MASK, 1)
    eg.run(nsamp, nburn, p, vg)
    c = eg.counters
    print("murray 4096x16: remote steps %d, passes %d" % (c["remote_steps"], c["remote_passes"]))
    assert c["remote_steps"] == eo.remote_steps and c["remote_passes"] == eo.remote_passes
    assert np.array_equal(eg.accept_mask, eo.accept_mask)
    for name in ("state", "mean", "var", "musigall"):
        assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name


def test_reference_accept_rates_at_full_size():
    """BASELINE.md's measured reference accept rates (fraction of (step >= 1, chain) rows that changed),
    same job shapes, on the GPU engine: 8-D x 4096 (200+50) -> 0.470; 16-D x 65 536 (100+20) -> 0.0468.
    Statistical (the RNG differs from MKL's by design)."""
    import mcpar_amd as M
    for d, n, nburn, nsamp, ref, tol in ((8, 4096, 200, 50, 0.470, 0.015), (16, 65536, 100, 20, 0.0468, 0.003)):
        vg, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
        eg = M.Engine(d, n, pl=1.0)
        eg.run(nsamp, nburn, O.default_pinit(d, n), vg)
        s = eg.samples.reshape(nsamp, n, d + 1)
        changed = np.any(s[1:, :, :d] != s[:-1, :, :d], axis=2)
        assert abs(changed.mean() - ref) < tol, (d, n, changed.mean())


@pytest.mark.parametrize("d,n,mix", [(16, 20480, 0), (16, 24576, 0), (16, 32768, 0), (8, 49152, 0), (32, 16384, 0), (32, 14336, 8), (16, 28672, 3)])
def test_small_n_mode_with_five_to_eight_owners_per_workgroup(d, n, mix):
    """k_run_small with 5-8 owner wavefronts per workgroup (no recorders, 8-11 generators): between 16 k and 32 k
    chains x 16-D the one-launch run beats the fused kernels' 1.5-2 waves per SIMD.  State, log-likelihood, moments,
    per-chain accept counts, tuner trace and sample rows against the oracle, every step of a short job."""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    p = O.default_pinit(d, n)
    if mix:  # a sum of `mix` unit Gaussians (the C5 likelihood): means 5k/(K-1) in every dimension, weights (5, 1, ..., 1)
        means = np.stack([np.full(d, 5.0 * k / (mix - 1)) for k in range(mix)]).astype(np.float32)
        params = np.concatenate([means.ravel(), [5.0] + [1.0] * (mix - 1)]).astype(np.float32)
        vo, _ko = O.make_vlfunc(O.VL_GAUSSMIX, d, params, mix)
        vg, _kg = M.make_vlfunc(M.VL_GAUSSMIX, d, params, mix)
    else:
        vo, _ko = O.make_vlfunc(O.VL_ROSENBROCK1, d)
        vg, _kg = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    eo.set_record(samples=True, mask=False)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_PERSIST, 1)
    eo.run(60, 110, p, vo)
    eg.run(60, 110, p, vg)
    c = eg.counters
    assert c["kernel_launches"] == 1, c  # the whole run was one k_run_small launch
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
    np.testing.assert_array_equal(eg.accept_counts, eo.accept_counts)
    np.testing.assert_array_equal(eg.tuner_trace.view(np.uint32), eo.tuner_trace.view(np.uint32))
    for name in ("state", "loglike", "mean", "var", "musigall", "samples"):
        a, b = getattr(eg, name), getattr(eo, name)
        assert a.shape == b.shape, name
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
