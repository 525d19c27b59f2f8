"""Randomised configurations (fixed seed): every combination of dimension, chain count, loop lengths,
sync period, proposal mix, likelihood, covariance, fusion mode, launch cap and thinning must match the
oracle bit for bit."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

VL_ROSENBROCK2_FIXED = 6  # MCX_VL_ROSENBROCK2_FIXED / MCXO_VL_ROSENBROCK2_FIXED (flagged variant, not the reference's)


def spd(rng, d):
    a = rng.normal(size=(d, d)).astype(np.float32)
    return (a @ a.T / d + 0.5 * np.eye(d)).astype(np.float32)


def one_case(rng, idx):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    kind = int(rng.choice([O.VL_ROSENBROCK1, O.VL_GAUSSIAN, O.VL_GAUSSMIX, O.VL_ROSENBROCK2, O.VL_DUALGAUSS, VL_ROSENBROCK2_FIXED]))
    d = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16, 20, 24, 32, 36, 48]))
    if kind == O.VL_ROSENBROCK1:
        d = max(2, d + (d & 1))
    if kind in (O.VL_ROSENBROCK2, VL_ROSENBROCK2_FIXED):
        d = max(2, d)
    if kind == O.VL_DUALGAUSS:
        d = 2
    n = int(rng.integers(1, 200)) if rng.random() < 0.9 else int(rng.integers(200, 3000))  # (several exclusion groups)
    nburn = int(rng.choice([0, 1, 49, 52, 60, 101, 130]))
    nsamp = int(rng.choice([0, 1, 5, 9, 10, 11, 37, 64]))
    pl = float(rng.choice([1.0, 0.9, 0.7, 0.4]))
    sync = int(rng.choice([1, 2, 7, 10, 50]))
    params, K = None, 0
    if kind == O.VL_GAUSSIAN and rng.random() < 0.7:
        params = np.concatenate([rng.normal(size=d), rng.uniform(0.3, 3, d)]).astype(np.float32)
    if kind == O.VL_GAUSSMIX:
        K = int(rng.choice([1, 2, 5, 8, 11]))
        params = np.concatenate([rng.normal(0, 2, K * d), rng.uniform(0.5, 4, K)]).astype(np.float32)
    if kind == O.VL_DUALGAUSS:
        params = [float(rng.uniform(0.5, 6))]
    incov = spd(rng, d) if rng.random() < 0.3 else None
    fuse = int(rng.random() < 0.8)
    mask = int(rng.random() < 0.5)
    maxseg = int(rng.choice([1, 3, 16, 256]))
    stride = int(rng.choice([1, 1, 2, 5]))
    persist = int(rng.choice([-1, -1, 0, 1]))   # small-n mode: one launch per run / per-segment kernels
    split = int(rng.choice([-1, -1, 0, 1]))     # (persist off:) pre-generated normals or generated in the step kernel
    sink = int(rng.choice([0, 0, 1, 3, 10, 64]))  # > 0: samples streamed through the sink in blocks of that many steps
    cull = int(rng.choice([-1, -1, 0, 1, 2, 3]))   # exact exclusion of far Gaussians in the Murray sweeps: auto / off / boxes / one direction / per-pair bound
    bpl = int(rng.choice([0, 0, 1, 2, 4]))      # parameter blocks per lane of the hot-path kernel (0 = automatic)
    # round 5's options, from a generator of their own (the cases of the fixed set stay the cases they were):
    rng5 = np.random.default_rng(1000003 * (idx + 1) + int(os.environ.get("MCX_FUZZ_SEED", "0")))
    async_run = int(rng5.random() < 0.3)         # mcx_run returns once the run is queued (runs that qualify; the getters finish it)
    overlap = int(rng5.choice([0, 0, 2, 4]))     # Murray passes over many chains by column chunks on two streams
    self_report = int(rng5.random() < 0.7)       # the run's last small-n launch tells the host itself / counters by copy
    if os.environ.get("MCX_FUZZ_CULL"):         # a soak of one screen: every case with it
        cull = int(os.environ["MCX_FUZZ_CULL"])
    desc = dict(idx=idx, kind=kind, d=d, n=n, nburn=nburn, nsamp=nsamp, pl=pl, sync=sync, K=K, fullcov=incov is not None,
                fuse=fuse, mask=mask, maxseg=maxseg, stride=stride, persist=persist, split=split, sink=sink, cull=cull, bpl=bpl,
                async_run=async_run, overlap=overlap, self_report=self_report)
    p = (rng.normal(0, 0.7, (n, d))).astype(np.float32)
    vo, k1 = O.make_vlfunc(kind, d, params, K)
    eo = O.Engine(d, n, pl=pl, sync=sync, threads=8 if n >= 200 else 1)
    eo.run(nsamp, nburn, p, vo, incov)
    vg, k2 = M.make_vlfunc(kind, d, params, K)
    eg = M.Engine(d, n, pl=pl, sync=sync)
    eg.set_option(E.OPT_ACCEPT_MASK, mask)
    eg.set_option(E.OPT_FUSE, fuse)
    eg.set_option(E.OPT_MAX_SEGMENT, maxseg)
    eg.set_option(E.OPT_SAMPLE_STRIDE, stride)
    eg.set_option(E.OPT_PERSIST, persist)
    eg.set_option(E.OPT_SPLIT_RNG, split)
    eg.set_option(E.OPT_CULL, cull)
    eg.set_option(E.OPT_BLOCKS_PER_LANE, bpl)
    eg.set_option(E.OPT_ASYNC_RUN, async_run)
    eg.set_option(E.OPT_MURRAY_OVERLAP, overlap)
    eg.set_option(E.OPT_SELF_REPORT, self_report)
    streamed, texts = [], []
    as_text = bool(sink) and rng.random() < 0.2 and n * nsamp * (d + 1) < 200000  # the blocks as text (mcx_set_text_sink)
    if as_text:
        eg.set_text_sink(lambda first, nsteps, text: texts.append((first, bytes(text))) and 0, sink)
    elif sink:
        eg.set_sink(lambda first, nsteps, rows: streamed.append((first, rows.copy())) and 0, sink)
    eg.run(nsamp, nburn, p, vg, incov)
    c = eg.counters
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main, desc
    assert c["remote_steps"] == eo.remote_steps and c["remote_passes"] == eo.remote_passes, desc
    if mask:
        assert np.array_equal(eg.accept_mask, eo.accept_mask), desc
    assert np.array_equal(eg.accept_counts, eo.accept_counts), desc
    assert np.array_equal(eg.tuner_trace.view(np.uint32), eo.tuner_trace.view(np.uint32)), desc
    names = ["state", "loglike", "chol"] + (["mean", "var", "musigall"] if nsamp > 0 else [])
    for name in names:
        a, b = getattr(eg, name), getattr(eo, name)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, desc)
    want = eo.samples.reshape(nsamp, n, d + 1)[::stride].reshape(-1, d + 1) if nsamp else np.zeros((0, d + 1), np.float32)
    if as_text:
        from test_gpu_text import libc_text
        assert [f for f, t in texts] == sorted(f for f, t in texts), desc
        assert b"".join(t for f, t in texts) == libc_text(want), ("text", desc)
        return
    if sink:
        assert [f for f, r in streamed] == sorted(f for f, r in streamed), desc
        got = np.concatenate([r for f, r in streamed]) if streamed else np.zeros((0, d + 1), np.float32)
        if nsamp:
            ll = want[:, d]
            ok = ll > -np.inf
            lm, pm = eg.maxlike()
            if ok.any():
                i = int(np.argmax(np.where(ok, ll, -np.inf)))
                assert lm == ll[i] and np.array_equal(pm.view(np.uint32), want[i, :d].view(np.uint32)), ("maxlike", desc)
    else:
        got = eg.samples
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), ("samples", desc)


def test_random_configurations():
    import os
    # MCX_FUZZ_SEED / MCX_FUZZ_CASES: a soak run with other seeds (the default is the fixed regression set)
    rng = np.random.default_rng(int(os.environ.get("MCX_FUZZ_SEED", "20261003")))
    for idx in range(int(os.environ.get("MCX_FUZZ_CASES", "400"))):
        one_case(rng, idx)


def test_random_multishard_configurations():
    from test_gpu_multishard import run_sharded_gpu
    import os
    rng = np.random.default_rng(int(os.environ.get("MCX_FUZZ_SEED", "77")))
    for idx in range(int(os.environ.get("MCX_FUZZ_SHARD_CASES", "24"))):
        d = int(rng.choice([2, 4, 8, 16, 20, 32]))
        n = int(rng.integers(3, 120))
        nshards = int(rng.choice([2, 3, 5]))
        nburn = int(rng.choice([0, 60, 110]))
        nsamp = int(rng.choice([9, 10, 21, 47]))
        pl = float(rng.choice([1.0, 0.8, 0.5]))
        sync = int(rng.choice([1, 3, 10]))
        eager = int(rng.random() < 0.5)
        mask = int(rng.random() < 0.5)
        desc = dict(idx=idx, d=d, n=n, nshards=nshards, nburn=nburn, nsamp=nsamp, pl=pl, sync=sync, eager=eager, mask=mask)
        vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
        eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl, sync=sync) for s in range(nshards)]
        O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
        egs = run_sharded_gpu(d, n, nshards, nburn, nsamp, pl, sync=sync, eager=eager, mask=mask)
        for s in range(nshards):
            c = egs[s].counters
            assert c["naccept_main"] == eos[s].naccept_main and c["remote_passes"] == eos[s].remote_passes, desc
            for name in ("state", "mean", "var", "samples", "musigall"):
                assert np.array_equal(getattr(egs[s], name).view(np.uint32), getattr(eos[s], name).view(np.uint32)), (name, s, desc)
