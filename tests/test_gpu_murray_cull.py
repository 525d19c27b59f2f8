"""Exact exclusion of far Gaussians in the Murray sweeps (mcx_remote.hpp, k_cull_*): sorting the active chains,
boxing every wavefront's 128 chains and skipping the Q_i that are provably too far from all of them must not
change a single bit -- with the exclusion forced on (MCX_OPT_CULL = 1: boxes; 2: one direction, mcx_cull_proj.hpp;
3: the per-pair bound on the matrix cores, mcx_screen.hpp), off
(0) and automatic (-1), against the
oracle, which knows nothing of it -- and must actually exclude most pairs on BASELINE-shaped states."""
import numpy as np
import pytest

import oracle_lib as O
from test_gpu_configs import THREADS, mix_params, same_bits

pytestmark = pytest.mark.gpu


def realistic_state(rng, N, d, n, stuck_every=0):
    ms = np.empty((N, d, 2), np.float32)
    ms[:, :, 0] = rng.normal(0.4, 0.4, (N, d))
    ms[:, :, 1] = rng.uniform(0.01, 0.12, (N, d)) ** 2
    if stuck_every:
        ms[::stuck_every, :, 1] = np.float32(1e-14) / 29
    pv = (ms[:n, :, 0] + np.sqrt(ms[:n, :, 1]) * rng.standard_normal((n, d))).astype(np.float32)
    if stuck_every:
        pv[::stuck_every] = ms[:n:stuck_every, :, 0]
    return ms, pv


@pytest.mark.parametrize("d,n,nshards", [(16, 1500, 1), (32, 700, 2), (16, 129, 3), (32, 4096, 1), (16, 1501, 1), (16, 999, 3),
                                         (16, 65, 5), (32, 333, 4), (16, 8200, 2)])
def test_gen_remote_same_bits_with_and_without_exclusion(d, n, nshards):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    rng = np.random.default_rng(d * 1000 + n)
    N = n * nshards
    shard = nshards - 1  # (own Gaussians at offset shard * n: the min-arg sweep's starting point)
    ms, pv = realistic_state(rng, N, d, n, stuck_every=7)
    own = slice(shard * n, (shard + 1) * n)
    pv = (ms[own, :, 0] + np.sqrt(ms[own, :, 1]) * rng.standard_normal((n, d))).astype(np.float32)
    pv[::7] = ms[own][::7, :, 0]
    eo = O.Engine(d, n, nshards=nshards, shard=shard, threads=THREADS)
    ro = eo.gen_remote(41, pv, ms)
    for mode in (1, 2, 3, 0, -1):  # boxes / one direction / per-pair bound (mcx_screen.hpp) / off / automatic
        eg = M.Engine(d, n, nshards=nshards, shard=shard)
        eg.set_option(E.OPT_CULL, mode)
        rg = eg.gen_remote(41, pv, ms)
        assert rg[4] == ro[4], mode
        for a, b, name in zip(rg[:4], ro[:4], ("ptrial", "cfac", "mutrial", "sigtrial")):
            assert same_bits(a, b), (mode, name)
        eg.close()


@pytest.mark.parametrize("cfg", ["rosen16", "mix32"])
def test_whole_murray_job_same_bits_and_most_pairs_excluded(cfg):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    if cfg == "rosen16":
        d, n, nburn, nsamp, pl = 16, 8192, 500, 60, 0.9
        spec_o, spec_g = (O.VL_ROSENBROCK1, d), (M.VL_ROSENBROCK1, d)
    else:
        d, n, nburn, nsamp, pl = 32, 4096, 500, 60, 0.9
        params = mix_params(d, 8)
        spec_o, spec_g = (O.VL_GAUSSMIX, d, params, 8), (M.VL_GAUSSMIX, d, params, 8)
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(*spec_o)
    eo = O.Engine(d, n, pl=pl, threads=THREADS)
    eo.set_record(samples=False, mask=False)
    eo.run(nsamp, nburn, p, vo)
    assert eo.remote_steps >= 3
    vg, k2 = M.make_vlfunc(*spec_g)
    frac = {}
    for mode in (1, 2, 3, 0):
        eg = M.Engine(d, n, pl=pl)
        eg.set_option(E.OPT_CULL, mode)
        eg.run(nsamp, nburn, p, vg)
        c = eg.counters
        assert c["remote_steps"] == eo.remote_steps and c["remote_passes"] == eo.remote_passes
        assert c["naccept_main"] == eo.naccept_main
        assert np.array_equal(eg.accept_counts, eo.accept_counts)
        for name in ("state", "loglike", "mean", "var", "musigall"):
            assert same_bits(getattr(eg, name), getattr(eo, name)), (mode, name)
        frac[mode] = c["remote_pairs_evaluated"] / float(c["remote_pairs"])
        eg.close()
    print("pairs left after the exclusion test: boxes %.3f, one direction %.3f, per-pair bound %.3f of all (%s)"
          % (frac[1], frac[2], frac[3], cfg))
    # (no screen: every pair the reference loops over, plus the proposals the late passes swept ahead of their turn)
    assert 1.0 <= frac[0] < 1.1 and frac[1] <= 1.05 and frac[2] <= 1.05
    if cfg == "rosen16":  # (the 32-D mixture: pairs are dead across the mixture's axis, which no bound for 128 chains sees
        assert frac[1] < 0.8 and frac[2] < 0.8  # -- the per-pair bound does)
    assert frac[3] < 0.6


def test_exclusion_keeps_nan_and_degenerate_inputs_identical():
    """a Gaussian with an enormous 1/sig2, chains far outside every box, a group of identical chains"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    rng = np.random.default_rng(99)
    d, n = 16, 640
    ms, pv = realistic_state(rng, n, d, n)
    ms[5, :, 1] = 1e-30
    ms[6, :, 0] = 1e6
    pv[10:150] = pv[10]
    pv[200] = 50.0
    eo = O.Engine(d, n, threads=THREADS)
    ro = eo.gen_remote(3, pv, ms)
    for mode in (1, 2, 3, 0):
        eg = M.Engine(d, n)
        eg.set_option(E.OPT_CULL, mode)
        rg = eg.gen_remote(3, pv, ms)
        assert rg[4] == ro[4]
        for a, b, name in zip(rg[:4], ro[:4], ("ptrial", "cfac", "mutrial", "sigtrial")):
            assert same_bits(a, b), (mode, name)
        eg.close()



@pytest.mark.parametrize("cfg,chunks", [("rosen16", 4), ("rosen16", 8), ("mix32", 2), ("mix32", 8)])
def test_big_passes_by_column_chunks_on_two_streams_same_bits(cfg, chunks):
    """MCX_OPT_MURRAY_OVERLAP: the Gaussians of a pass over many chains cut into column chunks, chunk c + 1 screened on the
    matrix cores while chunk c is swept on the vector units (src/mcpar.cc:367-395 is one loop over all Gaussians there; block
    partials are combined in index order whatever stream produced them): the whole job must equal the oracle bit for bit,
    pass counts included, and the chunked launches must really have happened"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    if cfg == "rosen16":
        d, n, nburn, nsamp, pl = 16, 16384, 300, 40, 0.85
        spec_o, spec_g = (O.VL_ROSENBROCK1, d), (M.VL_ROSENBROCK1, d)
    else:
        d, n, nburn, nsamp, pl = 32, 8192, 300, 40, 0.85
        par = mix_params(d, 8)
        spec_o, spec_g = (O.VL_GAUSSMIX, d, par, 8), (M.VL_GAUSSMIX, d, par, 8)
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(*spec_o)
    eo = O.Engine(d, n, pl=pl, threads=THREADS)
    eo.set_record(samples=False, mask=False)
    eo.run(nsamp, nburn, p, vo)
    assert eo.remote_steps > 0
    launches = {}
    for c in (0, chunks):
        vg, _k2 = M.make_vlfunc(*spec_g)
        eg = M.Engine(d, n, pl=pl)
        eg.set_option(E.OPT_SAMPLES, 0)
        eg.set_option(E.OPT_MURRAY_OVERLAP, c)
        eg.run(nsamp, nburn, p, vg)
        cn = eg.counters
        assert (cn["remote_steps"], cn["remote_passes"], cn["naccept_main"]) == (eo.remote_steps, eo.remote_passes, eo.naccept_main), c
        for name in ("state", "loglike", "mean", "var", "musigall"):
            assert same_bits(getattr(eg, name), getattr(eo, name)), (c, name)
        launches[c] = cn["kernel_launches"]
        eg.close()
    assert launches[chunks] > launches[0]  # two big passes per Murray step went out as 2 * chunks launches each
