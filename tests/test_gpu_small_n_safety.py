"""The one-launch small-n kernel (k_run_small) must fail, not hang: its tuner
meetings need every workgroup of the grid resident at once.  A meeting that cannot complete is abandoned after
MCX_OPT_MEET_TIMEOUT_MS, the launch writes nothing back, and mcx_run repeats the run on the per-segment
kernels -- same bits, and it says so in mcx_counters.meet_timeouts.  Also here: the Cholesky factor read back
after a one-launch run that follows a full-covariance run."""
import os
import stat
import time

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def oracle_run(d, n, nburn, nsamp, pl=1.0):
    vo, k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=pl, threads=8)
    eo.run(nsamp, nburn, O.default_pinit(d, n), vo)
    return eo


def test_meeting_that_cannot_complete_is_abandoned_and_the_run_repeated():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 8, 4096, 160, 40
    p = O.default_pinit(d, n)
    vo, k0 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    eo.set_record(samples=False, mask=False)
    vg, k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_PERSIST, 1)
    eg.set_option(E.OPT_SAMPLES, 0)

    def same_as_oracle():
        c = eg.counters
        assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
        assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
        for name in ("state", "loglike", "mean", "var"):
            assert same_bits(getattr(eg, name), getattr(eo, name)), name

    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, p, vg)  # the one-launch kernel as it normally runs
    assert eg.counters["meet_timeouts"] == 0 and eg.counters["kernel_launches"] <= 3
    same_as_oracle()
    # now every meeting waits for one workgroup more than the grid has: it can never complete
    eg.set_option(E.OPT_DEBUG_MEET, 1)
    eg.set_option(E.OPT_MEET_TIMEOUT_MS, 50)
    eo.run(nsamp, nburn, p, vo)  # (a second run continues the RNG step counter, on both sides)
    t0 = time.time()
    eg.run(nsamp, nburn, p, vg)
    dt = time.time() - t0
    assert dt < 1.0, "abandoning a meeting took %.2f s" % dt
    assert eg.counters["meet_timeouts"] == 1
    assert eg.counters["kernel_launches"] > 3  # the per-segment kernels did the job
    same_as_oracle()
    # the engine keeps to the per-segment kernels afterwards
    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, p, vg)
    assert eg.counters["meet_timeouts"] == 0 and eg.counters["kernel_launches"] > 3
    same_as_oracle()
    # switching the hook off gives the one-launch kernel back
    eg.set_option(E.OPT_DEBUG_MEET, 0)
    eo.run(nsamp, nburn, p, vo)
    eg.run(nsamp, nburn, p, vg)
    assert eg.counters["meet_timeouts"] == 0 and eg.counters["kernel_launches"] <= 3
    same_as_oracle()


def test_abandoned_meeting_with_a_sample_sink_delivers_only_good_rows():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp = 8, 2048, 120, 48
    eo = oracle_run(d, n, nburn, nsamp)
    vg, k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_PERSIST, 1)
    eg.set_option(E.OPT_DEBUG_MEET, 2)
    eg.set_option(E.OPT_MEET_TIMEOUT_MS, 30)
    got = []
    eg.set_sink(lambda first, ns, rows: got.append((first, rows.copy())) and 0, 16)
    eg.run(nsamp, nburn, O.default_pinit(d, n), vg)
    assert eg.counters["meet_timeouts"] == 1
    rows = np.concatenate([r for _, r in sorted(got, key=lambda t: t[0])])
    assert same_bits(rows, eo.samples)


def test_lock_file_is_at_a_fixed_path_whatever_tmpdir_says(tmp_path, monkeypatch):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    d, n = 8, 1024
    vg, k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_PERSIST, 1)
    eg.run(10, 60, O.default_pinit(d, n), vg)
    assert not any(f.startswith("mcx_meet_") for f in os.listdir(str(tmp_path)))
    locks = [f for f in os.listdir("/dev/shm") if f.startswith("mcx_meet_")]
    assert locks, "no lock file under /dev/shm"
    st = os.lstat(os.path.join("/dev/shm", locks[0]))
    assert stat.S_ISREG(st.st_mode)


def test_get_chol_after_full_covariance_run_then_one_launch_run():
    """the one-launch kernel writes back the diagonal of the factor only; a full factor left by an
    earlier run must not survive below it."""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n = 8, 2048
    rng = np.random.default_rng(3)
    a = rng.normal(size=(d, d))
    cov = (a @ a.T / d + np.eye(d)).astype(np.float32) * 0.01
    vg, k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    vo, k2 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    p = O.default_pinit(d, n)
    eg = M.Engine(d, n, pl=1.0)
    eg.set_option(E.OPT_PERSIST, 1)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    eg.run(20, 110, p, vg, incov=cov)
    eo.run(20, 110, p, vo, incov=cov)
    assert same_bits(eg.chol, eo.chol) and np.any(np.tril(eg.chol, -1) != 0)
    eg.run(20, 110, p, vg)  # identity factor: the one-launch kernel
    eo.run(20, 110, p, vo)
    assert same_bits(eg.chol, eo.chol)
    assert np.all(np.tril(eg.chol, -1) == 0)
    assert same_bits(eg.state, eo.state)
