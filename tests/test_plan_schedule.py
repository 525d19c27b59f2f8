"""Host logic of the engine without a GPU: mcx_plan() is the launch schedule mcx_run executes.
Checked against the reference's loop structure (src/mcpar.cc:55-97, 99-210) and the oracle's coin."""
import numpy as np
import pytest

import oracle_lib as O


def coin_remote(isamp, nburn, sync, pl, seed=8675309, tbase=0):
    if isamp < sync:
        return False
    t = tbase + nburn + isamp
    u = O.lib().mcxo_u24(O.philox([t, 0, 0, 0], [seed, 2])[0])
    return not (u <= np.float32(pl))


def get_plan(**kw):
    from mcpar_amd import engine as E
    return E.plan(**kw)


@pytest.mark.parametrize("nburn", [0, 1, 51, 52, 53, 100, 102, 500, 777])
def test_burn_in_segments_and_tuner_checks(nburn):
    p = get_plan(nsamp=0, nburn=nburn)
    segs = [it for it in p if it[0] == "burn_segment"]
    tun = [it for it in p if it[0] == "tuner"]
    assert len(segs) == len(tun)
    covered = []
    for (_, first, n, _a), (_, last, n2, check) in zip(segs, tun):
        covered += list(range(first, first + n))
        assert last == first + n - 1 and n2 == n
    assert covered == list(range(nburn))
    # the reference evaluates the rate at the end of iteration isamp when isamp > irate, irate = 50, 100, ...
    want = [s for s in range(nburn) if s > 50 and (s - 1) % 50 == 0]
    assert [last for (_, last, _n, check) in tun if check] == want


def test_max_segment_caps_launch_length():
    p = get_plan(nsamp=100, nburn=120, pl=1.0, max_segment=7)
    assert max(it[2] for it in p if it[0] in ("burn_segment", "main_segment")) == 7
    # a capped burn segment that ends before the check step must not trigger the tuner decision
    for it in p:
        if it[0] == "tuner" and it[3]:
            assert it[1] > 50 and (it[1] - 1) % 50 == 0


@pytest.mark.parametrize("pl,nshards,eager,fused", [(0.9, 1, 0, 1), (0.7, 2, 0, 1), (0.7, 2, 1, 1), (0.8, 4, 0, 0),
                                                    (1.0, 8, 0, 1), (1.0, 8, 1, 1), (0.5, 1, 0, 0)])
def test_main_loop_schedule(pl, nshards, eager, fused):
    nsamp, nburn, sync = 173, 60, 10
    p = get_plan(nsamp=nsamp, nburn=nburn, sync=sync, pl=pl, nshards=nshards, eager=eager, fused=fused)
    main = [it for it in p if it[0] in ("main_segment", "remote_step")]
    # every main-loop step exactly once, in order; Murray steps exactly where the coin says
    covered, remote = [], []
    for kind, first, n, aux in main:
        covered += list(range(first, first + n))
        if kind == "remote_step":
            remote.append(first)
    assert covered == list(range(nsamp))
    assert remote == [s for s in range(nsamp) if coin_remote(s, nburn, sync, pl)]
    names = [it[0] for it in p]
    if nshards == 1:
        assert "gather_begin" not in names and "gather_wait" not in names
        return
    begins = [it[1] for it in p if it[0] == "gather_begin"]
    sync_points = list(range(0, nsamp, sync))
    if eager:  # the reference's schedule: a gather at every sync point
        assert begins == sync_points
        for kind, first, n, aux in main:  # launches never cross a sync point
            if kind == "main_segment":
                assert aux == -1 and (first // sync == (first + n - 1) // sync)
    else:  # lazy: a gather only right before a Murray step that follows a new sync point, and one at the end
        assert len(begins) <= min(len(sync_points), len(remote) + 1)
        for b in begins:
            assert b in remote or b == nsamp
        # the snapshot a gather ships is the one of the last sync point <= that step
        published = 0
        for kind, first, n, aux in p:
            if kind == "publish":
                published = first
            elif kind == "main_segment" and aux >= 0:
                assert fused and (first + aux + 1) % sync == 0 and first < first + aux + 1 < first + n
                assert (first + n - 1) // sync * sync == first + aux + 1  # the LAST sync point inside
                published = first + aux + 1
            elif kind == "gather_begin":
                assert published == (first if first < nsamp else nsamp - 1) // sync * sync
    # a Murray step always sees a completed gather and a current own slot
    for i, it in enumerate(p):
        if it[0] == "remote_step":
            assert p[i - 1] == ("publish", it[1], 0, 0) and p[i - 2][0] == "gather_wait"
    assert p[-1] == ("publish", nsamp, 0, 0)


def test_output_dump_points_follow_the_reference():
    for nsamp in (8, 50, 51, 100, 1000):
        p = get_plan(nsamp=nsamp, nburn=0, pl=1.0, has_output_hook=1, max_segment=4096)
        outstep = nsamp // 10 if nsamp > 50 else 5  # src/mcpar.cc:110
        assert [it[1] for it in p if it[0] == "output"] == [s for s in range(1, nsamp) if s % outstep == 0]
        for kind, first, n, aux in p:
            if kind == "main_segment":
                assert first // outstep == (first + n - 1) // outstep
        # without a hook nothing forces a launch boundary
        q = get_plan(nsamp=nsamp, nburn=0, pl=1.0, has_output_hook=0, max_segment=4096)
        assert [it for it in q if it[0] == "main_segment"] == [("main_segment", 0, nsamp, -1)]


def test_second_run_uses_the_advanced_step_counter():
    a = get_plan(nsamp=200, nburn=50, pl=0.8, tbase=0)
    b = get_plan(nsamp=200, nburn=50, pl=0.8, tbase=250)
    ra = [it[1] for it in a if it[0] == "remote_step"]
    rb = [it[1] for it in b if it[0] == "remote_step"]
    assert ra != rb
    assert rb == [s for s in range(200) if coin_remote(s, 50, 10, 0.8, tbase=250)]


def test_sink_blocks_cut_the_main_loop():
    """mcx_set_sink: the main loop is cut at multiples of the block length, every block is handed over once, in
    order, the last (partial) one at the end; no launch straddles a block"""
    for nsamp, block, pl in ((100, 10, 1.0), (95, 25, 0.8), (7, 10, 1.0), (40, 1, 0.9), (256, 64, 1.0)):
        p = get_plan(nsamp=nsamp, nburn=60, pl=pl, sink_block=block, max_segment=4096)
        sinks = [(f, n) for k, f, n, a in p if k == "sink"]
        want = [(min(b + block, nsamp), min(block, nsamp - b)) for b in range(0, nsamp, block)]
        assert sinks == want, (nsamp, block, sinks)
        done = 0
        for k, f, n, a in p:
            if k in ("main_segment", "remote_step"):
                assert f == done and f // block == (f + n - 1) // block  # inside one block
                done += n
            if k == "sink":
                assert f == done
        assert done == nsamp
        q = get_plan(nsamp=nsamp, nburn=60, pl=pl, sink_block=0, max_segment=4096)
        assert not any(k == "sink" for k, *_ in q)
