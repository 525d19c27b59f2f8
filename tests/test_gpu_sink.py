"""Streaming sample sink (mcx_set_sink): the role of MCout as src/mcpar.cc:176-182 fills it and :110-119 dumps it,
without keeping the whole run in HBM -- a ring of four blocks on the device, rows staged out on a second stream.
The rows a sink receives must be exactly the rows of the whole-run store, the running maximum-likelihood sample
the first strict maximum in (step, chain) order (src/mcout.cc:140-144)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def first_max(rows, d):
    ll = rows[:, d]
    ok = ll > -np.inf
    if not ok.any():
        return -np.inf, np.zeros(d, np.float32)
    i = int(np.argmax(np.where(ok, ll, -np.inf)))  # argmax returns the first of equal maxima
    return ll[i], rows[i, :d]


@pytest.mark.parametrize("d,n,nburn,nsamp,pl,block,stride,persist", [
    (16, 512, 120, 103, 0.85, 10, 1, -1),   # small-n one-launch kernel, Murray steps in between, ragged last block
    (16, 512, 120, 103, 0.85, 10, 2, -1),   # thinned
    (8, 4096, 60, 64, 1.0, 7, 1, 0),        # per-segment kernels (pre-generated normals)
    (16, 20000, 55, 40, 1.0, 16, 1, -1),    # hot-path fused kernel, more blocks than ring slots
    (16, 20000, 55, 45, 0.9, 6, 3, -1),
    (6, 100, 30, 25, 0.8, 1, 1, -1),        # generic kernel (d % 4 != 0), one step per block
    (16, 300, 0, 9, 1.0, 100, 1, -1),       # block longer than the run, no burn-in
])
def test_sink_rows_equal_the_store(d, n, nburn, nsamp, pl, block, stride, persist):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    p = O.default_pinit(d, n)
    kind = M.VL_ROSENBROCK1
    vg, keep = M.make_vlfunc(kind, d)
    ref = M.Engine(d, n, pl=pl)
    ref.set_option(E.OPT_SAMPLE_STRIDE, stride)
    ref.set_option(E.OPT_PERSIST, persist)
    ref.run(nsamp, nburn, p, vg)
    want = ref.samples
    wl, wp = ref.maxlike()
    el, ep = first_max(want, d)
    assert wl == el and same_bits(wp, ep)  # device arg-max == first strict maximum
    eng = M.Engine(d, n, pl=pl)
    eng.set_option(E.OPT_SAMPLE_STRIDE, stride)
    eng.set_option(E.OPT_PERSIST, persist)
    got, calls = [], []

    def sink(first, nsteps, rows):
        calls.append((first, nsteps))
        got.append(rows.copy())
        return 0
    eng.set_sink(sink, block)
    eng.run(nsamp, nburn, p, vg)
    rows = np.concatenate(got)
    kb = (block + stride - 1) // stride  # block rounded up to a multiple of the stride, in kept steps
    nkeep = (nsamp + stride - 1) // stride
    assert calls == [(f, min(kb, nkeep - f)) for f in range(0, nkeep, kb)]
    assert same_bits(rows, want)
    gl, gp = eng.maxlike()
    assert gl == wl and same_bits(gp, wp)
    for name in ("state", "mean", "var", "loglike"):
        assert same_bits(getattr(eng, name), getattr(ref, name)), name
    with pytest.raises(M.McxError):
        eng.samples_range(0, 1)  # the run was streamed: no whole-run store
    # a second run on the same engine streams again (RNG continues), and removing the sink restores the store
    got.clear(); calls.clear()
    eng.run(nsamp, nburn, p, vg)
    ref.run(nsamp, nburn, p, vg)
    assert same_bits(np.concatenate(got), ref.samples)
    eng.set_sink(None, 0)
    eng.run(nsamp, nburn, p, vg)
    ref.run(nsamp, nburn, p, vg)
    assert same_bits(eng.samples, ref.samples)


@pytest.mark.parametrize("d,n,nburn,nsamp,pl,block,stride,persist", [
    (16, 512, 120, 103, 0.85, 10, 1, -1),   # small-n one-launch kernel, Murray steps in between, ragged last block
    (16, 512, 120, 103, 0.85, 10, 2, -1),   # thinned
    (8, 4096, 60, 64, 1.0, 7, 1, 0),        # per-segment kernels (pre-generated normals)
    (16, 20000, 55, 40, 1.0, 16, 1, -1),    # hot-path fused kernel, more blocks than ring slots
    (16, 20000, 55, 45, 0.9, 6, 3, -1),
    (6, 100, 30, 25, 0.8, 1, 1, -1),        # generic kernel (d % 4 != 0), one step per block
    (16, 300, 0, 9, 1.0, 100, 1, -1),       # block longer than the run, no burn-in
    (16, 8192, 110, 64, 1.0, 16, 1, -1),    # one-launch kernel without recorders (two owner wavefronts per workgroup)
])
def test_sink_rows_equal_the_oracle(d, n, nburn, nsamp, pl, block, stride, persist):
    """The streamed rows against the CPU oracle's own sample store (not against the HIP engine's): the rows MCout
    would hold after src/mcpar.cc:176-182, and its running maximum (src/mcout.cc:140-144) -- over the whole matrix of
    kernel paths, block lengths and thinning that test_sink_rows_equal_the_store covers."""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    p = O.default_pinit(d, n)
    vo, keep_o = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=pl, threads=8)
    eo.set_record(samples=True, mask=False, stride=stride)
    eo.run(nsamp, nburn, p, vo)
    want = eo.samples
    vg, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng = M.Engine(d, n, pl=pl)
    eng.set_option(E.OPT_SAMPLE_STRIDE, stride)
    eng.set_option(E.OPT_PERSIST, persist)
    got = []

    def sink(first, nsteps, rows):
        got.append(rows.copy())
        return 0
    eng.set_sink(sink, block)
    eng.run(nsamp, nburn, p, vg)
    rows = np.concatenate(got)
    assert rows.shape == want.shape
    assert same_bits(rows, want)
    gl, gp = eng.maxlike()
    el, ep = first_max(want, d)
    assert gl == el and same_bits(gp, ep)
    eo.close()


def test_sink_failure_is_reported():
    import mcpar_amd as M
    d, n = 8, 256
    eng = M.Engine(d, n, pl=1.0)
    eng.set_sink(lambda first, nsteps, rows: 1, 5)
    with pytest.raises(M.McxError):
        eng.run(20, 10, O.default_pinit(d, n), M.make_vlfunc(M.VL_ROSENBROCK1, d)[0])


def test_maxlike_ignores_nan_and_minus_inf():
    """a likelihood that returns NaN / -inf for some chains: neither can be the maximum (src/mcout.cc:140: `>`)"""
    import mcpar_amd as M
    d, n = 2, 64

    def lik(x):
        y = -0.5 * (x ** 2).sum(1)
        y[::3] = np.nan
        y[1::3] = -np.inf
        return y.astype(np.float32)
    v, keep = M.make_vlfunc(M.VL_HOST, d, host_fn=lik)
    eng = M.Engine(d, n, pl=1.0)
    eng.run(12, 8, O.default_pinit(d, n), v)
    rows = eng.samples
    gl, gp = eng.maxlike()
    el, ep = first_max(rows, d)
    assert gl == el and same_bits(gp, ep) and np.isfinite(gl)
