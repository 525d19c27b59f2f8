"""k_fused_fastb (mcx_fastb.hpp): the hot-path kernel with two or four 4-parameter blocks per lane must give the
bits of the one-block kernel -- i.e. of the oracle -- for every likelihood of the hot path, ragged chains
(d = 12, 20: a lane with a live and a dead block), thinned sample stores and the in-kernel exchange snapshot."""
import numpy as np
import pytest

import oracle_lib as O
from test_gpu_configs import mix_params, same_bits

pytestmark = pytest.mark.gpu


def specs(kind, d):
    import mcpar_amd as M
    if kind == "rosen":
        return (O.VL_ROSENBROCK1, d), (M.VL_ROSENBROCK1, d)
    if kind == "gauss":
        p = np.concatenate([np.linspace(-1, 1, d), np.linspace(0.5, 2.0, d)]).astype(np.float32)
        return (O.VL_GAUSSIAN, d, p), (M.VL_GAUSSIAN, d, p)
    p = mix_params(d, 8)
    return (O.VL_GAUSSMIX, d, p, 8), (M.VL_GAUSSMIX, d, p, 8)


@pytest.mark.parametrize("bpl", [2, 4])
@pytest.mark.parametrize("kind,d,n", [("rosen", 16, 3000), ("rosen", 8, 1111), ("rosen", 12, 700), ("rosen", 32, 640),
                                      ("rosen", 20, 513), ("gauss", 16, 900), ("gauss", 24, 300), ("mix", 32, 512),
                                      ("mix", 8, 777)])
def test_blocks_per_lane_same_bits_as_oracle(kind, d, n, bpl):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    so, sg = specs(kind, d)
    nburn, nsamp = 130, 70
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(*so)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    eo.run(nsamp, nburn, p, vo)
    vg, k2 = M.make_vlfunc(*sg)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
    eg.set_option(E.OPT_PERSIST, 0)
    eg.set_option(E.OPT_SPLIT_RNG, 0)  # the plain hot-path kernel, not the small-n modes
    eg.run(nsamp, nburn, p, vg)
    c = eg.counters
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
    assert np.array_equal(eg.accept_counts, eo.accept_counts)
    assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
    for name in ("state", "loglike", "mean", "var", "musigall", "samples"):
        assert same_bits(getattr(eg, name), getattr(eo, name)), name
    eg.close()


@pytest.mark.parametrize("bpl", [2, 4])
def test_blocks_per_lane_thinned_store_and_second_run(bpl):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp, stride = 16, 2048, 60, 45, 4
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    eo.set_record(samples=True, mask=False, stride=stride)
    vg, k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    for o, v in ((E.OPT_BLOCKS_PER_LANE, bpl), (E.OPT_PERSIST, 0), (E.OPT_SPLIT_RNG, 0), (E.OPT_SAMPLE_STRIDE, stride)):
        eg.set_option(o, v)
    for rep in range(2):
        eo.run(nsamp, nburn, p, vo)
        eg.run(nsamp, nburn, p, vg)
        kept = (nsamp + stride - 1) // stride
        assert same_bits(eg.samples, eo.samples[-kept * n:]), rep
        for name in ("state", "mean", "var"):
            assert same_bits(getattr(eg, name), getattr(eo, name)), (rep, name)


@pytest.mark.parametrize("bpl", [2, 4])
def test_blocks_per_lane_multishard_snapshot(bpl):
    """two shards on one GPU, lazy schedule: fused segments span sync points and the kernel snapshots the slot"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    from test_gpu_configs import run_sharded_async
    d, n, nshards, nburn, nsamp, pl = 16, 2048, 2, 100, 55, 0.93
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl, threads=8) for s in range(nshards)]
    O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)

    def setup(s, e):
        e.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
        e.set_option(E.OPT_PERSIST, 0)
        e.set_option(E.OPT_SPLIT_RNG, 0)
    egs, nbegin = run_sharded_async(d, n, nshards, nburn, nsamp, pl, 0, setup=setup)
    for s in range(nshards):
        assert egs[s].counters["remote_passes"] == eos[s].remote_passes
        for name in ("state", "mean", "var", "samples", "musigall"):
            assert same_bits(getattr(egs[s], name), getattr(eos[s], name)), (s, name)


# ---- the one-launch small-n kernel (k_run_small<LPC2, BPL, ...>, mcx_persist.hpp) with two / four blocks per lane -------
@pytest.mark.parametrize("bpl", [2, 4])
@pytest.mark.parametrize("kind,d,n", [("rosen", 16, 3000), ("rosen", 8, 1111), ("rosen", 12, 700), ("rosen", 32, 640),
                                      ("rosen", 24, 300), ("gauss", 16, 900), ("gauss", 24, 300), ("mix", 32, 512),
                                      ("mix", 8, 777), ("mix", 16, 2100)])
def test_small_n_blocks_per_lane_same_bits_as_oracle(kind, d, n, bpl):
    """forced blocks per lane (illegal combinations -- d = 12 with two, d = 8 with four -- fall back to one)"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    so, sg = specs(kind, d)
    nburn, nsamp = 130, 70
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(*so)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    vg, k2 = M.make_vlfunc(*sg)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
    eg.set_option(E.OPT_PERSIST, 1)
    for rep in range(2):  # the second run continues the step counter and starts from staged moments
        eo.run(nsamp, nburn, p, vo)
        eg.run(nsamp, nburn, p, vg)
        c = eg.counters
        assert c["kernel_launches"] <= 3 and c["meet_timeouts"] == 0, c
        assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
        assert np.array_equal(eg.accept_counts, eo.accept_counts)
        assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
        for name in ("state", "loglike", "mean", "var", "musigall", "chol"):
            assert same_bits(getattr(eg, name), getattr(eo, name)), (rep, name)
        assert same_bits(eg.samples, eo.samples[-nsamp * n:]), rep  # (the oracle appends a run's rows to its store)
    eg.close()


@pytest.mark.parametrize("d,n,want_bpl", [(16, 8192, 1), (16, 16384, 2), (8, 4096, 1), (8, 16384, 1), (16, 12288, 1)])
def test_small_n_automatic_blocks_per_lane_full_job(d, n, want_bpl):
    """the automatic choice (mcxk_persist_bpl: two blocks per lane from four owner wavefronts per workgroup on, where the
    halved grid still fills the chip) on 8192 x 16-D -- the per-GPU shape of a strong-scaled C3 job: two owners, no
    recorders, 32-step phases -- and its neighbours; the whole R-local job (nburn 500 + nsamp 1000, thinned store)
    against the oracle"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    nburn, nsamp, stride = 500, 1000, 50
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    eo.set_record(samples=True, mask=False, stride=stride)
    eo.run(nsamp, nburn, p, vo)
    vg, k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_SAMPLE_STRIDE, stride)
    eg.run(nsamp, nburn, p, vg)
    c = eg.counters
    assert c["kernel_launches"] <= 3 and c["meet_timeouts"] == 0, c
    assert c["small_n_blocks_per_lane"] == want_bpl, c
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
    assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
    kept = (nsamp + stride - 1) // stride
    assert same_bits(eg.samples, eo.samples[-kept * n:])
    for name in ("state", "loglike", "mean", "var", "musigall"):
        assert same_bits(getattr(eg, name), getattr(eo, name)), name
    eg.close()


@pytest.mark.parametrize("bpl", [2, 4])
def test_small_n_blocks_per_lane_multishard_snapshot(bpl):
    """two shards on one GPU, lazy schedule, one launch per stretch: the recorder wavefronts snapshot the slot of both blocks"""
    from mcpar_amd import engine as E
    from test_gpu_configs import run_sharded_async
    d, n, nshards, nburn, nsamp, pl = 16, 2048, 2, 100, 55, 0.93
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl, threads=8) for s in range(nshards)]
    O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)

    def setup(s, e):
        e.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
        e.set_option(E.OPT_PERSIST, 1)
    egs, nbegin = run_sharded_async(d, n, nshards, nburn, nsamp, pl, 0, setup=setup)
    for s in range(nshards):
        assert egs[s].counters["remote_passes"] == eos[s].remote_passes
        for name in ("state", "mean", "var", "samples", "musigall"):
            assert same_bits(getattr(egs[s], name), getattr(eos[s], name)), (s, name)


@pytest.mark.parametrize("bpl", [1, 2, 0], ids=["one-block", "mirrored", "engine-choice"])
@pytest.mark.parametrize("kind,d,n,pl", [("rosen", 32, 640, 1.0), ("rosen", 16, 1500, 0.9), ("gauss", 32, 300, 0.9), ("mix", 32, 512, 1.0),
                                         ("rosen", 28, 257, 1.0), ("rosen", 12, 700, 0.9), ("gauss", 16, 900, 1.0), ("mix", 16, 333, 0.85)])
def test_full_covariance_kernels_same_bits_as_oracle(kind, d, n, pl, bpl):
    """Full-covariance proposals (src/mcpar.cc:302-312 with covar_setup's factor, :454-484) on both hot-path kernels: one block
    per lane (k_fused_fast<..., FULL>: the factor in registers up to 16-D, in LDS above) and two MIRRORED blocks per lane
    (k_fused_fastb<..., FULL>, mcx_fastb.hpp: lane q2 holds blocks q2 and NB - 1 - q2, the first block's columns above the
    diagonal are never issued) -- ragged chains (d = 28, 12: a lane whose second / first block is absent), thinned stores and
    a sharded snapshot included.  Same bits as the oracle whichever kernel runs."""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    so, sg = specs(kind, d)
    nburn, nsamp = 110, 45
    a = np.random.default_rng(5 + d).normal(size=(d, d))
    cov = (0.02 * (np.eye(d) + 0.5 * a @ a.T / d)).astype(np.float32)
    p = O.default_pinit(d, n)
    vo, k1 = O.make_vlfunc(*so)
    eo = O.Engine(d, n, pl=pl, threads=8)
    eo.run(nsamp, nburn, p, vo, cov)
    vg, k2 = M.make_vlfunc(*sg)
    eg = M.Engine(d, n, pl=pl)
    eg.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
    eg.set_option(E.OPT_PERSIST, 0)
    eg.set_option(E.OPT_SPLIT_RNG, 0)
    eg.run(nsamp, nburn, p, vg, cov)
    c = eg.counters
    assert (c["naccept_burn"], c["naccept_main"]) == (eo.naccept_burn, eo.naccept_main)
    assert (c["remote_steps"], c["remote_passes"]) == (eo.remote_steps, eo.remote_passes)
    assert np.array_equal(eg.accept_counts, eo.accept_counts)
    assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
    for name in ("state", "loglike", "mean", "var", "musigall", "samples", "chol"):
        assert same_bits(getattr(eg, name), getattr(eo, name)), name
    # a second run on the same engine (the factor is reinstalled, the tuner starts over), thinned store
    eo.set_record(samples=True, mask=False, stride=3)
    eo.run(nsamp, nburn, p, vo, cov)
    eg.set_option(E.OPT_SAMPLE_STRIDE, 3)
    eg.run(nsamp, nburn, p, vg, cov)
    for name in ("state", "mean", "var"):
        assert same_bits(getattr(eg, name), getattr(eo, name)), name
    so_ = eo.samples
    assert same_bits(eg.samples, so_[so_.shape[0] - eg.samples.shape[0]:])
    eg.close()


def test_full_covariance_engine_choice_at_the_headline_size():
    """16-D full covariance at 65 536 chains: the engine's own choice of kernel (the mirrored one from that size on since
    round 5, one block per lane below) -- a short job, all bits against the oracle."""
    import mcpar_amd as M
    d, n, nburn, nsamp = 16, 65536, 60, 25
    a = np.random.default_rng(77).normal(size=(d, d))
    cov = (0.02 * (np.eye(d) + 0.5 * a @ a.T / d)).astype(np.float32)
    p = O.default_pinit(d, n)
    vo, _k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=16)
    eo.run(nsamp, nburn, p, vo, cov)
    vg, _k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.run(nsamp, nburn, p, vg, cov)
    c = eg.counters
    assert (c["naccept_burn"], c["naccept_main"]) == (eo.naccept_burn, eo.naccept_main)
    assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
    for name in ("state", "loglike", "mean", "var", "musigall", "samples", "chol"):
        assert same_bits(getattr(eg, name), getattr(eo, name)), name
    eg.close(); eo.close()
