"""MCX_OPT_ASYNC_RUN: mcx_run returns once the run (MCPar::run, src/mcpar.cc:17-214) is queued; whoever looks next -- a getter,
mcx_get_counters, mcx_synchronize -- finishes it first, so the results are a synchronous run's, bit for bit; runs queued
back to back without looking overlap their launch and completion latencies (at most two in flight), and only the last
one's results are ever seen, as when each is waited for.  A run whose tuner meeting was abandoned is still repeated --
at the point where somebody looks."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def check(eg, eo, what):
    c = eg.counters  # (finishes the run)
    assert (c["naccept_burn"], c["naccept_main"]) == (eo.naccept_burn, eo.naccept_main), what
    assert same_bits(eg.tuner_trace, eo.tuner_trace), what
    for name in ("state", "loglike", "mean", "var", "musigall"):
        assert same_bits(getattr(eg, name), getattr(eo, name)), (what, name)
    a, b = eg.samples, eo.samples
    assert same_bits(a, b[b.shape[0] - a.shape[0]:]), (what, "samples")


@pytest.mark.parametrize("d,n,host_pinit", [(16, 8192, False), (8, 4096, True), (16, 65536, False), (16, 700, True)],
                         ids=["one-launch-8192", "one-launch-4096-host-pinit", "segments-65536", "tiny-host-pinit"])
@pytest.mark.parametrize("self_report", [1, 0], ids=["kernel-reports", "counters-by-copy"])
def test_async_runs_equal_synchronous_ones(d, n, host_pinit, self_report):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    nburn, nsamp = (500, 300) if n > 10000 else (160, 90)
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=16)
    vg, _k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_ASYNC_RUN, 1)
    eg.set_option(E.OPT_SELF_REPORT, self_report)
    if not host_pinit:
        eg.stage_pinit(p)
    for r in range(3):  # every run looked at
        eo.run(nsamp, nburn, p, vo)
        eg.run(nsamp, nburn, p if host_pinit else None, vg)
        check(eg, eo, "looked at, run %d" % r)
    for r in range(5):  # queued back to back, only the last looked at (the oracle makes all five: the step counter moves on)
        eo.run(nsamp, nburn, p, vo)
        eg.run(nsamp, nburn, p if host_pinit else None, vg)
    check(eg, eo, "back to back")
    # switching the option off finishes what is in flight; a Murray job (pl < 1) is waited for whatever the option says
    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, p if host_pinit else None, vg)
    eg.set_option(E.OPT_ASYNC_RUN, 0)
    check(eg, eo, "option off")
    eg.close(); eo.close()


@pytest.mark.parametrize("self_report", [1, 0], ids=["kernel-reports", "counters-by-copy"])
def test_an_abandoned_meeting_in_an_async_run_is_repeated_when_somebody_looks(capfd, self_report):
    """MCX_OPT_DEBUG_MEET makes every meeting wait for a workgroup that does not exist: the one-launch kernel gives up after
    MCX_OPT_MEET_TIMEOUT_MS, the host learns of it when the run is finished -- and repeats it on the per-segment kernels,
    from the state the run started from (caller memory that has long been changed), same bits, counters say so"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 16, 4096, 160, 60
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    vg, _k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_ASYNC_RUN, 1)
    eg.set_option(E.OPT_SELF_REPORT, self_report)
    eg.set_option(E.OPT_MEET_TIMEOUT_MS, 5)
    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, p, vg)
    check(eg, eo, "healthy")
    assert eg.counters["small_n_launches"] == 1 and eg.counters["meet_timeouts"] == 0
    eg.set_option(12, 1)  # MCX_OPT_DEBUG_MEET: one arrival too many expected
    mine = p.copy()
    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, mine, vg)
    mine[:] = 99.0  # the caller's memory is the caller's again once mcx_run has returned
    check(eg, eo, "abandoned, repeated")
    c = eg.counters
    assert c["meet_timeouts"] == 1 and c["meet_timeouts_total"] == 1 and c["small_n_launches"] == 0
    # an abandoned run that nobody looked at: nothing to repeat, but the books know, and the engine keeps off the kernel
    eg.set_option(12, 0)   # (also lets the one-launch kernel in again)
    eg.set_option(12, 1)
    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, p, vg)   # abandoned in flight ...
    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, p, vg)   # ... overtaken by this one (abandoned too, repeated when looked at)
    check(eg, eo, "overtaken")
    assert eg.counters["meet_timeouts_total"] >= 2
    eg.close(); eo.close()


@pytest.mark.parametrize("d,n,nburn,nsamp", [(16, 8192, 160, 90), (8, 4096, 0, 50), (8, 1000, 120, 0), (32, 2048, 100, 40)])
def test_the_last_launch_reports_the_end_of_a_synchronous_run_itself(d, n, nburn, nsamp):
    """MCX_OPT_SELF_REPORT (default on): the one-launch kernel that ends a run writes the counters and a serial number to
    pinned host memory and mcx_run spins on it instead of queueing a copy and sleeping -- the same run with the option off
    and the oracle's: same bits, same counters, run after run (the serial numbers move on, the slots turn)."""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=16)
    vg, _k2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    engines = []
    for sr in (1, 0):
        eg = M.Engine(d, n, pl=1.0)
        eg.set_option(E.OPT_SELF_REPORT, sr)
        engines.append(eg)
    for r in range(6):
        eo.run(nsamp, nburn, p, vo)
        for sr, eg in zip((1, 0), engines):
            eg.run(nsamp, nburn, p, vg)
            c = eg.counters
            assert (c["naccept_burn"], c["naccept_main"]) == (eo.naccept_burn, eo.naccept_main), (sr, r)
            assert c["small_n_launches"] == 1 and c["meet_timeouts"] == 0, (sr, r, c)
            assert same_bits(eg.state, eo.state) and same_bits(eg.tuner_trace, eo.tuner_trace), (sr, r)
            if nsamp:
                assert same_bits(eg.var, eo.var) and same_bits(eg.musigall, eo.musigall), (sr, r)
    for eg in engines:
        eg.close()
    eo.close()
