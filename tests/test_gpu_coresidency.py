"""The one-launch small-n kernel's tuner meetings need every workgroup of its grid resident at once.  What happens when
a FOREIGN kernel holds part of the chip: tests/cpp/occupy.hip keeps M compute units busy -- one
workgroup per CU, 120 KB of LDS each, so no workgroup of k_run_small (136 KB) fits beside it -- for a set time.

  * shorter than MCX_OPT_MEET_TIMEOUT_MS: the first meeting simply lasts until the foreign kernel is gone;
  * longer: the meeting is abandoned, the run repeated on the per-segment kernels (which need no co-residency and run
    beside the foreign kernel), and the engine says so: meet_timeouts, meet_timeouts_total, MCX_VERBOSE;
  * the engine's OWN gather in flight (MCX_OPT_ASYNC_TAIL): a launch with meetings starts behind it (default), so the
    question does not arise; with MCX_OPT_MEET_UNDER_GATHER = 1 it starts under it and the timeout is the net.
Results equal the oracle's in every case."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def occupy(tmp_path_factory):
    import mcpar_amd as M
    M.load()
    so = str(tmp_path_factory.mktemp("occupy") / "liboccupy.so")
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-O2",
                           os.path.join(ROOT, "tests", "cpp", "occupy.hip"), "-o", so])
    lib = C.CDLL(so)
    lib.occupy_start.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p)]
    lib.occupy_done.argtypes = [C.c_void_p]
    lib.occupy_wait.argtypes = [C.c_void_p]
    return lib


def start(lib, nblocks, ms, lds=120 * 1024, threads=256):
    h = C.c_void_p()
    assert lib.occupy_start(nblocks, threads, lds, ms, C.byref(h)) == 0
    time.sleep(0.002)  # its workgroups are on the chip
    return h


D, N, NBURN, NSAMP = 16, 8192, 160, 40  # 256 owner wavefronts with two blocks per lane: one workgroup on every CU


def engines():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    vo, k0 = O.make_vlfunc(O.VL_ROSENBROCK1, D)
    eo = O.Engine(D, N, pl=1.0, threads=8)
    eo.set_record(samples=False, mask=False)
    vg, k1 = M.make_vlfunc(M.VL_ROSENBROCK1, D)
    eg = M.Engine(D, N, pl=1.0)
    eg.set_option(E.OPT_PERSIST, 1)
    eg.set_option(E.OPT_SAMPLES, 0)
    return eo, vo, eg, vg, (k0, k1)


def check(eg, eo):
    c = eg.counters
    assert c["naccept_burn"] == eo.naccept_burn and c["naccept_main"] == eo.naccept_main
    assert np.array_equal(eg.tuner_trace, eo.tuner_trace)
    for name in ("state", "loglike", "mean", "var"):
        assert same_bits(getattr(eg, name), getattr(eo, name)), name


def test_meeting_outlasts_a_short_foreign_kernel(occupy):
    from mcpar_amd import engine as E
    import mcpar_amd as M
    ncu = M.device_info()[1]
    eo, vo, eg, vg, keep = engines()
    p = O.default_pinit(D, N)
    eo.run(NSAMP, NBURN, p, vo)
    eg.run(NSAMP, NBURN, p, vg)  # warm: lock file, code objects
    assert eg.counters["small_n_launches"] >= 1
    check(eg, eo)
    eg.set_option(E.OPT_MEET_TIMEOUT_MS, 400)
    h = start(occupy, ncu // 2, 40.0)
    t0 = time.perf_counter()
    eo.run(NSAMP, NBURN, p, vo)
    eg.run(NSAMP, NBURN, p, vg)
    dt = time.perf_counter() - t0
    still = not occupy.occupy_done(h)
    assert occupy.occupy_wait(h) == 0
    c = eg.counters
    assert c["meet_timeouts"] == 0 and c["small_n_launches"] >= 1, c
    check(eg, eo)
    # (the run could not have finished while half the CUs were taken: it ended with or after the foreign kernel)
    assert not still or dt > 0.02, (still, dt)
    eg.close()


def test_long_foreign_kernel_abandons_the_meeting_loudly_and_the_run_is_repeated(occupy, capfd):
    from mcpar_amd import engine as E
    import mcpar_amd as M
    ncu = M.device_info()[1]
    os.environ["MCX_VERBOSE"] = "1"
    try:
        eo, vo, eg, vg, keep = engines()
        p = O.default_pinit(D, N)
        eg.set_option(E.OPT_MEET_TIMEOUT_MS, 15)
        h = start(occupy, ncu // 2, 300.0)
        t0 = time.perf_counter()
        eo.run(NSAMP, NBURN, p, vo)
        eg.run(NSAMP, NBURN, p, vg)
        dt = time.perf_counter() - t0
        c = eg.counters
        assert c["meet_timeouts"] == 1 and c["meet_timeouts_total"] == 1 and c["small_n_launches"] == 0, c
        check(eg, eo)
        assert "abandoned" in capfd.readouterr().err
        # the engine keeps to the per-segment kernels (no second timeout while the foreign kernel is still there) ...
        eo.run(NSAMP, NBURN, p, vo)
        eg.run(NSAMP, NBURN, p, vg)
        c = eg.counters
        assert c["meet_timeouts"] == 0 and c["meet_timeouts_total"] == 1 and c["small_n_launches"] == 0, c
        check(eg, eo)
        assert occupy.occupy_wait(h) == 0
        # ... and tries the one-launch kernel again after 16 runs
        for _ in range(16):
            eo.run(NSAMP, NBURN, p, vo)
            eg.run(NSAMP, NBURN, p, vg)
        c = eg.counters
        assert c["small_n_launches"] >= 1 and c["meet_timeouts"] == 0 and c["meet_timeouts_total"] == 1, c
        check(eg, eo)
        eg.close()
        assert dt < 5.0
    finally:
        del os.environ["MCX_VERBOSE"]


@pytest.mark.parametrize("under", [0, 1], ids=["behind-the-gather", "under-the-gather"])
def test_own_gather_in_flight_and_a_launch_with_meetings(occupy, under):
    """two shards on one GPU, MCX_OPT_ASYNC_TAIL = 2; the "gather" is a hook that, in BEGIN, also puts a 60 ms foreign
    kernel on half the CUs (a collective kernel waiting for a slow peer) and, in WAIT, waits for it.  Default: run 2's
    launch with meetings starts behind the gather -- no timeout although the meetings' limit is 15 ms.  With
    MCX_OPT_MEET_UNDER_GATHER = 1 it starts under it, is abandoned, and the run repeated: same bits either way."""
    import threading
    import mcpar_amd as M
    from mcpar_amd import engine as E
    ncu = M.device_info()[1]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    d, n, nshards, nburn, nsamp, pl = 16, 4096, 2, 110, 30, 1.0
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl, threads=8) for s in range(nshards)]
    for _ in range(2):
        O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    engs = [M.Engine(d, n, nshards=nshards, shard=s, pl=pl) for s in range(nshards)]
    vl, keep2 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    ptrs, held = [None] * nshards, [None] * nshards
    bar = threading.Barrier(nshards)
    errs = []

    def make_hook(s):
        def hook(phase, ptr, slot, shard, ns, stream):
            if phase == E.XCHG_BEGIN:
                # (both shards' earlier launches are over before the foreign kernel starts: the other shard's launch
                # with meetings must not find it in its way by being a little behind -- it only waits for its OWN gather)
                hip.hipDeviceSynchronize()
                bar.wait(timeout=60)
                if s == 0:  # (one foreign kernel per gather is enough: a quarter of the chip for 60 ms)
                    held[s] = start(occupy, ncu // 4, 60.0)
                return 0
            ptrs[s] = ptr
            hip.hipDeviceSynchronize()
            bar.wait(timeout=60)
            for r in range(ns):
                if r != s:
                    assert hip.hipMemcpy(ptr + r * slot * 4, ptrs[r] + r * slot * 4, slot * 4, 3) == 0
            hip.hipDeviceSynchronize()
            if held[s] is not None:
                assert occupy.occupy_wait(held[s]) == 0
                held[s] = None
            bar.wait(timeout=60)
            return 0
        return hook

    def work(s):
        try:
            e = engs[s]
            # 2 x 4096 x 16-D with one block per lane: 2 x 256 owner wavefronts, a workgroup on every CU for each engine
            for k, v in ((E.OPT_ASYNC_TAIL, 2), (E.OPT_PERSIST, 1), (E.OPT_BLOCKS_PER_LANE, 1), (E.OPT_MEET_TIMEOUT_MS, 15),
                         (E.OPT_MEET_UNDER_GATHER, under), (E.OPT_SAMPLES, 0)):
                e.set_option(k, v)
            e.set_exchange(make_hook(s))
            for _ in range(2):
                e.run(nsamp, nburn, O.default_pinit(d, n, g0=s * n), vl)
            e.synchronize()  # the last gather's WAIT, in step with the other shard's thread
        except Exception as ex:  # pragma: no cover
            errs.append(ex)
            bar.abort()

    th = [threading.Thread(target=work, args=(s,)) for s in range(nshards)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    total = sum(e.counters["meet_timeouts_total"] for e in engs)
    if not under:
        assert total == 0, [e.counters for e in engs]
    for s in range(nshards):
        for name in ("state", "mean", "var", "musigall"):
            assert same_bits(getattr(engs[s], name), getattr(eos[s], name)), (s, name)
        engs[s].close()
