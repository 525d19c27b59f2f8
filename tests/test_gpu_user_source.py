"""MCX_VL_SOURCE: a user's likelihood handed over as HIP source of device functions and compiled INTO the fused step
kernels (mcx_user.hip) -- the fast path for the reference's whole plug-in surface, VLFunc (src/vlfunc.hh:9-12, called at
src/mcpar.cc:60,160).  Parity: a source that restates Rosenbrock1 (src/rosenbrock.cc:4-21) must equal the built-in and
the CPU oracle bit for bit -- accept indices, state, moments, sample rows, tuner trace, Murray pass counts -- in both
forms (per-block partials / whole vector), on every kernel it can land on (hot-path, full covariance, generic), for a
whole C3 job; a likelihood with no built-in counterpart is checked against the oracle driven by a numpy functor."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src(name):
    return open(os.path.join(ROOT, "mcpar_amd", "examples", name)).read()


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def assert_equal_runs(eo, eg, what, mask=True):
    if mask:
        bad = np.argwhere(eo.accept_mask != eg.accept_mask)
        assert bad.size == 0, "%s: accept mask differs first at (step, chain) %s" % (what, bad[:3])
    c = eg.counters
    assert (c["naccept_burn"], c["naccept_main"]) == (eo.naccept_burn, eo.naccept_main), what
    assert (c["remote_steps"], c["remote_passes"]) == (eo.remote_steps, eo.remote_passes), what
    assert same_bits(eg.tuner_trace, eo.tuner_trace), what
    for name in ("state", "loglike", "mean", "var", "musigall", "samples"):
        a, b = getattr(eg, name), getattr(eo, name)
        if name == "samples":
            b = b[b.shape[0] - a.shape[0]:]
        assert same_bits(a, b), (what, name)


FORMS = [("user_rosenbrock1_blocks.hip", None), ("user_rosenbrock1_whole.hip", np.array([1.0], np.float32))]


@pytest.mark.parametrize("form", [0, 1], ids=["block-form", "whole-vector"])
@pytest.mark.parametrize("d,n,pl,mask,fullcov", [
    (16, 4096, 1.0, 0, 0),    # hot-path kernel (k_fused_fast's body), tuner + moments + rows inside the launches
    (16, 640, 0.8, 0, 0),     # + Murray steps: the user's eval kernel between genRemote and accept
    (8, 512, 0.85, 1, 0),     # accept mask -> the generic kernel
    (12, 300, 0.9, 0, 0),     # a lane without parameters (d = 12 on 4 lanes)
    (6, 257, 0.9, 0, 0),      # d % 4 != 0 -> generic kernel, a two-parameter last block
    (2, 1000, 0.9, 0, 0),     # one lane per chain
    (32, 512, 0.9, 0, 0),     # 8 lanes per chain
    (40, 200, 0.9, 0, 0),     # 16 lanes per chain: generic kernel only
    (16, 1024, 0.9, 0, 1),    # full covariance: the hot-path kernel's FULL form
    (32, 512, 1.0, 0, 1),
])
def test_restated_rosenbrock1_equals_builtin_and_oracle(form, d, n, pl, mask, fullcov):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    name, par = FORMS[form]
    nburn, nsamp = 160, 70
    p = O.default_pinit(d, n)
    incov = None
    if fullcov:
        a = np.random.default_rng(7 + d).normal(size=(d, d))
        incov = (0.02 * (np.eye(d) + 0.4 * a @ a.T / d)).astype(np.float32)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=pl, threads=4)
    eo.run(nsamp, nburn, p, vo, incov)
    vg, _k2 = M.make_vlfunc(M.VL_SOURCE, d, params=par, source=src(name))
    eg = M.Engine(d, n, pl=pl)
    eg.set_option(E.OPT_ACCEPT_MASK, mask)
    eg.run(nsamp, nburn, p, vg, incov)
    assert_equal_runs(eo, eg, "%s d=%d" % (name, d), mask=bool(mask))
    # the launch count says the steps were fused: a handful per run, not three per step
    launches = eg.counters["kernel_launches"]
    assert launches < 40 + 12 * eg.counters["remote_passes"] + 8 * eg.counters["remote_steps"], launches
    # the built-in on the same engine settings: same bits again (and a second run on the cached code object)
    vb, _k3 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eb = M.Engine(d, n, pl=pl)
    eb.run(nsamp, nburn, p, vb, incov)
    for nm in ("state", "mean", "var", "samples"):
        assert same_bits(getattr(eg, nm), getattr(eb, nm)), nm
    eo.run(nsamp, nburn, p, vo, incov)
    eg.run(nsamp, nburn, p, vg, incov)
    assert_equal_runs(eo, eg, "%s d=%d second run" % (name, d), mask=bool(mask))
    eg.close(); eb.close(); eo.close()


@pytest.mark.parametrize("form", [0, 1], ids=["block-form", "whole-vector"])
def test_whole_c3_job_with_a_user_source_is_bit_exact(form):
    """BASELINE config 3 as benchmarked (Rosenbrock1(16) x 65 536, nburn 500 + nsamp 1000, pl = 1) with the likelihood
    supplied as SOURCE: every accept decision (98.3 M, via the per-chain counts and the totals), the final state, the
    moments and strided sample rows equal the oracle's"""
    import mcpar_amd as M
    name, par = FORMS[form]
    d, n, nburn, nsamp = 16, 65536, 500, 1000
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=16)
    eo.set_record(samples=True, mask=False)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(M.VL_SOURCE, d, params=par, source=src(name))
    eg = M.Engine(d, n, pl=1.0)
    eg.run(nsamp, nburn, p, vg)
    c = eg.counters
    assert (c["naccept_burn"], c["naccept_main"]) == (eo.naccept_burn, eo.naccept_main)
    assert c["kernel_launches"] <= 20  # the built-in takes 15
    np.testing.assert_array_equal(eg.accept_counts, eo.accept_counts)
    assert same_bits(eg.tuner_trace, eo.tuner_trace)
    for nm in ("state", "loglike", "mean", "var"):
        assert same_bits(getattr(eg, nm), getattr(eo, nm)), nm
    so = eo.samples.reshape(nsamp, n, d + 1)
    for s in (0, 1, 499, 999):
        assert same_bits(eg.samples_range(s, 1), so[s]), s
    eg.close(); eo.close()


def banana_numpy(par):
    b, w0, w = (np.float32(v) for v in par)

    def fn(x):
        x = np.asarray(x, np.float32)
        y1 = (x[:, 1] + (b * x[:, 0]) * x[:, 0]) - np.float32(100.0) * b
        acc = (x[:, 0] * x[:, 0]) * w0
        acc = acc + (y1 * y1) * w
        for k in range(2, x.shape[1]):
            acc = acc + (x[:, k] * x[:, k]) * w
        return (np.float32(-0.5) * acc).astype(np.float32)
    return fn


@pytest.mark.parametrize("d,n,pl", [(4, 2048, 1.0), (8, 1000, 0.85), (2, 512, 0.9)])
def test_a_likelihood_without_builtin_counterpart_equals_the_oracle_with_a_host_functor(d, n, pl):
    import mcpar_amd as M
    par = np.array([0.03, 1.0 / 100.0, 1.0], np.float32)
    fn = banana_numpy(par)
    nburn, nsamp = 150, 60
    p = O.default_pinit(d, n)

    def tramp(ctx, npset, x, y):
        xa = np.ctypeslib.as_array(x, shape=(npset, d))
        np.ctypeslib.as_array(y, shape=(npset,))[:] = fn(xa)
        return 0
    cb = O.HOSTFN(tramp)
    vo, _k = O.make_vlfunc(O.VL_HOST, d, fn=cb)
    eo = O.Engine(d, n, pl=pl)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(M.VL_SOURCE, d, params=par, source=src("user_banana.hip"))
    eg = M.Engine(d, n, pl=pl)
    eg.run(nsamp, nburn, p, vg)
    assert_equal_runs(eo, eg, "banana d=%d" % d, mask=False)
    # the functor call itself (mcx_vlfunc_eval) on the same source
    x = np.random.default_rng(3).normal(size=(777, d)).astype(np.float32) * 3
    y = np.empty(777, np.float32)
    M._lib.check(M.load().mcx_vlfunc_eval(C.byref(vg), 777, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float))))
    assert same_bits(y, fn(x))
    eg.close(); eo.close()


def test_errors_are_loud():
    import mcpar_amd as M
    d, n = 8, 64
    p = O.default_pinit(d, n)
    eg = M.Engine(d, n, pl=1.0)
    bad, _k = M.make_vlfunc(M.VL_SOURCE, d, source="__device__ float mcx_user_loglike(const float *x, int d, const float *par) { return oops; }")
    with pytest.raises(M.McxError) as ei:
        eg.run(5, 5, p, bad)
    assert ei.value.code == 7 and "oops" in str(ei.value)
    empty = M._lib.VLFunc(M.VL_SOURCE, d, 0, None, M._lib.HOSTFN(), None)
    with pytest.raises(M.McxError):
        eg.run(5, 5, p, empty)
    # the engine is still usable
    vb, _k3 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eg.run(5, 5, p, vb)
    eg.close()


def test_a_whole_kernel_from_source_for_the_device_kind():
    """mcx_user_kernel_compile: the MCX_VL_DEVICE contract without hipcc at hand"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    text = open(os.path.join(ROOT, "tests", "cpp", "user_vlfunc_kernel.hip")).read()
    fn = E.compile_user_kernel(text, "user_rosenbrock8")
    d, n, nburn, nsamp = 8, 300, 120, 40
    p = O.default_pinit(d, n)
    vo, _k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=0.85)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(M.VL_DEVICE, d, device_fn=fn)
    eg = M.Engine(d, n, pl=0.85)
    eg.run(nsamp, nburn, p, vg)
    assert_equal_runs(eo, eg, "device kernel from source", mask=False)
    with pytest.raises(M.McxError):
        E.compile_user_kernel(text, "no_such_kernel")
    eg.close(); eo.close()


@pytest.mark.parametrize("d,n", [(8, 4096), (16, 8192), (16, 16384), (2, 1000), (32, 8192), (12, 2048)])
def test_block_form_source_takes_the_one_launch_small_n_kernel(d, n):
    """Few chains: burn-in with its tuner meetings, the start of the moments and the main loop are ONE launch of k_run_small
    (mcx_persist.hpp) -- also around a user's source in block form (built by hiprtc when the first run qualifies).  Same bits
    as the oracle and as the built-in; the counters say which kernel ran."""
    import mcpar_amd as M
    nburn, nsamp = 160, 90
    p = O.default_pinit(d, n)
    vo, _k = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=1.0, threads=8)
    vg, _k2 = M.make_vlfunc(M.VL_SOURCE, d, source=src("user_rosenbrock1_blocks.hip"))
    eg = M.Engine(d, n, pl=1.0)
    vb, _k3 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eb = M.Engine(d, n, pl=1.0)
    for r in range(2):
        eo.run(nsamp, nburn, p, vo)
        eg.run(nsamp, nburn, p, vg)
        assert_equal_runs(eo, eg, "small-n d=%d n=%d run %d" % (d, n, r), mask=False)
    eb.run(nsamp, nburn, p, vb)
    cg, cb = eg.counters, eb.counters
    assert cg["small_n_launches"] == cb["small_n_launches"] and cg["kernel_launches"] == cb["kernel_launches"], (cg, cb)
    if d % 4 == 0:
        assert cg["small_n_launches"] >= 1 and cg["kernel_launches"] <= 3
    # the whole-vector form has no small-n kernel: per-segment kernels, same bits
    vw, _k4 = M.make_vlfunc(M.VL_SOURCE, d, params=np.array([1.0], np.float32), source=src("user_rosenbrock1_whole.hip"))
    ew = M.Engine(d, n, pl=1.0)
    ew.run(nsamp, nburn, p, vw)
    eo2 = O.Engine(d, n, pl=1.0, threads=8)
    eo2.run(nsamp, nburn, p, vo)
    assert_equal_runs(eo2, ew, "whole-vector, few chains", mask=False)
    assert ew.counters["small_n_launches"] == 0
    for e in (eg, eb, ew, eo, eo2):
        e.close()


def test_the_cache_of_compiled_sources_lets_go_of_unused_entries(monkeypatch):
    """Code objects are cached per (device, np, text).  A host that writes constants into the text makes an entry per value:
    past the cap, entries no engine holds are unloaded -- and a text seen again simply compiles again, same bits."""
    import mcpar_amd as M
    monkeypatch.setenv("MCX_USER_CACHE_MAX", "2")
    d, n = 8, 256
    p = O.default_pinit(d, n)
    base = src("user_rosenbrock1_blocks.hip")
    states = []
    for k in (0, 1, 2, 3, 0):
        vg, _k = M.make_vlfunc(M.VL_SOURCE, d, source=base + "\n// variant %d\n" % k)
        eg = M.Engine(d, n, pl=1.0)
        eg.run(30, 60, p, vg)
        states.append(eg.state.copy())
        eg.close()
    for s in states[1:]:
        assert same_bits(s, states[0])
    # an entry an engine still holds survives the sweep: the engine keeps running on it
    va, _k = M.make_vlfunc(M.VL_SOURCE, d, source=base + "\n// held\n")
    ea = M.Engine(d, n, pl=1.0)
    vb, _k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eb = M.Engine(d, n, pl=1.0)
    ea.run(30, 60, p, va); eb.run(30, 60, p, vb)
    assert same_bits(ea.state, states[0]) and same_bits(eb.state, states[0])
    for k in (10, 11, 12):
        vg, _k = M.make_vlfunc(M.VL_SOURCE, d, source=base + "\n// variant %d\n" % k)
        eg = M.Engine(d, n, pl=1.0)
        eg.run(5, 0, p, vg)
        eg.close()
    ea.run(30, 60, p, va); eb.run(30, 60, p, vb)  # (a second run draws on: compared with the built-in's second run)
    assert same_bits(ea.state, eb.state)
    ea.close(); eb.close()
