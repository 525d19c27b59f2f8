"""N > 1 product path on ONE GPU: several engines (one per shard) run concurrently in threads and
exchange their musigall slots through the exchange hook (device-to-device copies standing in for
the RCCL all-gather); results must equal the oracle's in-process multi-shard run bit for bit."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def run_sharded_gpu(d, n, nshards, nburn, nsamp, pl, sync=10, eager=0, mask=1, vlspec=None, opts=None, runs=1, deferred=False):
    """deferred: the hook does nothing but note the request in BEGIN and moves the data in WAIT -- a gather that is truly
    in flight between the two: an engine that rewrites its slot, or reads the peers', before its WAIT gets wrong bits"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    engs = [M.Engine(d, n, nshards=nshards, shard=s, pl=pl, sync=sync) for s in range(nshards)]
    vl, keep = M.make_vlfunc(*(vlspec or (M.VL_ROSENBROCK1, d)))
    ptrs = [None] * nshards
    bar = threading.Barrier(nshards)
    errs = []

    def make_hook(s):
        def hook(phase, ptr, slot, shard, ns, stream):
            if deferred and phase == E.XCHG_BEGIN:
                return 0
            if not deferred and phase != E.XCHG_BEGIN:
                return 0
            ptrs[s] = ptr
            hip.hipDeviceSynchronize()   # every shard's slot is published
            bar.wait(timeout=60)
            for r in range(ns):          # pull every peer's slot (all-gather, in place)
                if r != s:
                    off = r * slot * 4
                    rc = hip.hipMemcpy(ptr + off, ptrs[r] + off, slot * 4, 3)
                    assert rc == 0
            hip.hipDeviceSynchronize()
            bar.wait(timeout=60)
            return 0
        return hook

    def work(s):
        try:
            engs[s].set_option(E.OPT_ACCEPT_MASK, mask)
            engs[s].set_option(E.OPT_EAGER_EXCHANGE, eager)
            for k, v in (opts or {}).items():
                engs[s].set_option(k, v)
            engs[s].set_exchange(make_hook(s))
            for _ in range(runs):
                engs[s].run(nsamp, nburn, O.default_pinit(d, n, g0=s * n), vl)
            if deferred:  # a gather left in flight moves its data in WAIT: every shard's thread must get there together
                engs[s].synchronize()
        except Exception as ex:  # pragma: no cover
            errs.append(ex)
            bar.abort()

    th = [threading.Thread(target=work, args=(s,)) for s in range(nshards)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    return engs


@pytest.mark.parametrize("eager", [0, 1], ids=["lazy", "eager"])
@pytest.mark.parametrize("nshards,pl", [(2, 0.7), (3, 0.8), (2, 1.0), (2, 0.93)])
def test_multishard_equals_oracle(nshards, pl, eager):
    """eager = the reference's schedule (a gather at every sync point); lazy = only the gathers a
    Murray step (or the end of the run) will read.  Both must equal the oracle bit for bit."""
    d, n, nburn, nsamp = 16, 96, 120, 60
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl) for s in range(nshards)]
    O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    egs = run_sharded_gpu(d, n, nshards, nburn, nsamp, pl, eager=eager)
    for s in range(nshards):
        eo, eg = eos[s], egs[s]
        assert np.array_equal(eg.accept_mask, eo.accept_mask), "shard %d" % s
        c = eg.counters
        assert c["remote_steps"] == eo.remote_steps and c["remote_passes"] == eo.remote_passes
        if eager:
            assert c["exchanges"] == nsamp // 10
        else:
            assert 1 <= c["exchanges"] <= min(nsamp // 10, c["remote_steps"] + 1)
        for name in ("state", "mean", "var", "samples"):
            assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name
        if pl < 1.0:
            assert c["remote_steps"] > 0
        # own slot is current, peers' slots are as of the last exchange (src/mcpar.cc:127-140,205-208)
        assert np.array_equal(eg.musigall.view(np.uint32), eo.musigall.view(np.uint32))


@pytest.mark.parametrize("deferred", [False, True], ids=["copy-in-begin", "copy-in-wait"])
@pytest.mark.parametrize("persist", [1, 2, 0], ids=["one-launch", "one-launch-under-gather", "segments"])
@pytest.mark.parametrize("eager", [0, 1], ids=["lazy", "eager"])
@pytest.mark.parametrize("pl,runs", [(0.7, 1), (1.0, 1), (0.8, 2), (1.0, 3)])
def test_last_gather_left_in_flight(pl, runs, eager, persist, deferred):
    """MCX_OPT_ASYNC_TAIL: mcx_run returns without waiting for the run's last gather and without the slot's final publish
    behind it; the getters (and the next run) finish them.  2 = with any exchange hook -- the default, 1, does it for
    the library's own RCCL exchange only, which takes several GPUs."""
    from mcpar_amd import engine as E
    d, n, nshards, nburn, nsamp = 16, 96, 2, 60, 50
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl) for s in range(nshards)]
    for _ in range(runs):
        O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    # (one-launch mode: a run that starts under the last run's gather waits for it before a launch with tuner meetings --
    # or, MCX_OPT_MEET_UNDER_GATHER = 1, launches its burn-in first, on its own, under the gather)
    egs = run_sharded_gpu(d, n, nshards, nburn, nsamp, pl, eager=eager, runs=runs, deferred=deferred,
                          opts={E.OPT_ASYNC_TAIL: 2, E.OPT_PERSIST: 1 if persist else 0, E.OPT_MEET_UNDER_GATHER: 1 if persist == 2 else 0})
    if runs > 1 and persist == 2:
        assert egs[0].counters["kernel_launches"] >= 2
    if runs > 1:
        assert egs[0].counters["exchange_waits"] >= 1
    for s in range(nshards):
        eo, eg = eos[s], egs[s]
        assert np.array_equal(eg.accept_mask, eo.accept_mask), "shard %d" % s
        # (first the peers' slots as of the last exchange and the own slot as of the last step: finish_tail)
        assert np.array_equal(eg.musigall.view(np.uint32), eo.musigall.view(np.uint32))
        for name in ("state", "mean", "var"):
            assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name
        # (the oracle appends a run's rows to its store, the engine keeps the last run's)
        assert np.array_equal(eg.samples.view(np.uint32), eo.samples[-nsamp * n:].view(np.uint32))
        eg.synchronize()
        assert np.array_equal(eg.musigall.view(np.uint32), eo.musigall.view(np.uint32))


def test_missing_exchange_hook_is_an_error():
    import mcpar_amd as M
    e = M.Engine(4, 8, nshards=2, shard=0)
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, 4)
    with pytest.raises(M.McxError) as ei:
        e.run(5, 5, np.zeros((8, 4), np.float32), vl)
    assert ei.value.code == 6


@pytest.mark.parametrize("eager", [0, 1], ids=["lazy", "eager"])
@pytest.mark.parametrize("pl", [0.8, 1.0])
def test_multishard_hot_path_kernel(pl, eager):
    """without accept-mask recording the hot-path kernel runs (incl. its in-kernel exchange snapshot)"""
    d, n, nshards, nburn, nsamp = 16, 160, 2, 120, 75
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl) for s in range(nshards)]
    O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    egs = run_sharded_gpu(d, n, nshards, nburn, nsamp, pl, eager=eager, mask=0)
    for s in range(nshards):
        eo, eg = eos[s], egs[s]
        c = eg.counters
        assert c["naccept_main"] == eo.naccept_main and c["remote_passes"] == eo.remote_passes
        for name in ("state", "mean", "var", "samples", "musigall"):
            assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("mode", ["pregen", "persistent"])
def test_small_n_modes_two_shards_long_segments(mode, stride):
    """Two shards of 8192 x 16-D chains, 600 local main-loop steps: in MCX_OPT_SPLIT_RNG mode a segment is longer
    than one chunk of pre-generated normals (256 steps here), so the launch loop rebases step index, sample rows
    and the in-kernel exchange snapshot per chunk; in MCX_OPT_PERSIST mode the whole run is one launch whose
    recorder wavefronts take the snapshot.  Thinned and unthinned sample store."""
    from mcpar_amd import engine as E
    d, n, nshards, nburn, nsamp, pl = 16, 8192, 2, 60, 600, 1.0
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl, threads=16) for s in range(nshards)]
    O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    opts = {E.OPT_SPLIT_RNG: 1, E.OPT_PERSIST: 1 if mode == "persistent" else 0, E.OPT_SAMPLE_STRIDE: stride,
            E.OPT_MAX_SEGMENT: 1 << 20}
    egs = run_sharded_gpu(d, n, nshards, nburn, nsamp, pl, mask=0, opts=opts)
    for s in range(nshards):
        eo, eg = eos[s], egs[s]
        assert eg.counters["naccept_main"] == eo.naccept_main and eg.counters["naccept_burn"] == eo.naccept_burn
        for name in ("state", "mean", "var", "musigall"):
            assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), (s, name)
        want = eo.samples.reshape(nsamp, n, d + 1)[::stride].reshape(-1, d + 1)
        assert np.array_equal(eg.samples.view(np.uint32), want.view(np.uint32)), s


def test_c5_shape_mixture_four_shards():
    """BASELINE config 5 in miniature: 32-D 8-component mixture, chains over 4 shards, Murray swaps"""
    d, K, n, nshards, nburn, nsamp, pl = 32, 8, 256, 4, 120, 60, 0.85
    means = np.stack([np.full(d, 5.0 * k / (K - 1)) for k in range(K)]).astype(np.float32)
    params = np.concatenate([means.ravel(), [5] + [1] * (K - 1)]).astype(np.float32)
    vo, keep = O.make_vlfunc(O.VL_GAUSSMIX, d, params, K)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl, threads=8) for s in range(nshards)]
    O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    assert eos[0].remote_steps > 0
    egs = run_sharded_gpu(d, n, nshards, nburn, nsamp, pl, mask=0, vlspec=(O.VL_GAUSSMIX, d, params, K))
    for s in range(nshards):
        c = egs[s].counters
        assert c["remote_passes"] == eos[s].remote_passes and c["naccept_main"] == eos[s].naccept_main
        for name in ("state", "mean", "var", "samples", "musigall"):
            assert np.array_equal(getattr(egs[s], name).view(np.uint32), getattr(eos[s], name).view(np.uint32)), name


def test_exchange_self_check_sees_every_slot_and_a_broken_exchange():
    """mcx_debug_fill_slot + mcx_debug_exchange + mcx_get_musigall: the start-up check bench.py (and any launcher) runs
    on a freshly made exchange before trusting it -- every shard fills its slot with shard + 1, one gather, slot r must
    be full of r + 1 everywhere.  A hook that forgets one peer must be caught."""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    d, n, nshards = 8, 300, 4
    for forget in (None, 2):
        engs = [M.Engine(d, n, nshards=nshards, shard=s) for s in range(nshards)]
        ptrs = [None] * nshards
        bar = threading.Barrier(nshards)
        res = [None] * nshards

        def make_hook(s):
            def hook(phase, ptr, slot, shard, ns, stream):
                if phase != E.XCHG_BEGIN:
                    return 0
                ptrs[s] = ptr
                hip.hipDeviceSynchronize()
                bar.wait(timeout=60)
                for r in range(ns):
                    if r != s and not (s == 0 and r == forget):  # shard 0 "forgets" peer `forget`
                        hip.hipMemcpy(ptr + r * slot * 4, ptrs[r] + r * slot * 4, slot * 4, 3)
                bar.wait(timeout=60)
                return 0
            return hook

        def work(s):
            engs[s].set_exchange(make_hook(s))
            res[s] = engs[s].exchange_self_check()

        th = [threading.Thread(target=work, args=(s,)) for s in range(nshards)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert res == ([True] * nshards if forget is None else [False] + [True] * (nshards - 1)), (forget, res)
        for e in engs:
            e.close()
