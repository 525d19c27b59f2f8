"""Sample rows as text on the GPU (mcx_samples_text / mcx_format_rows, mcx_text.hpp): byte for byte what the reference's
MCout::output prints (src/mcout.cc:41-45: `ostream << float` with the stream defaults = the C library's printf("%g"), two
blanks behind every field, a newline behind a row's last column)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

_libc = C.CDLL(None)
_libc.snprintf.restype = C.c_int
_libc.snprintf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_double]


def libc_text(rows):
    buf = C.create_string_buffer(64)
    out = []
    for row in rows:
        for v in row:
            _libc.snprintf(buf, 64, b"%g", C.c_double(float(v)))
            out.append(buf.value + b"  ")
        out.append(b"\n")
    return b"".join(out)


def test_every_kind_of_float_prints_like_the_c_library():
    from mcpar_amd import engine as E
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2 ** 32, 40000, dtype=np.uint64).astype(np.uint32)
    special = np.array([0, 0x80000000, 1, 0x80000001, 0x007fffff, 0x00800000, 0x7f7fffff, 0xff7fffff, 0x7f800000, 0xff800000,
                        0x7fc00000, 0xffc00000, 0x3f800000, 0x49742400, 0x49742408, 0x497423f8, 0x3dcccccd, 0x38d1b717,
                        0x38d1b716, 0x4b800000, 0x501502f9], np.uint32)  # +-0, denormals, limits, inf, nan, 1, ~1e6, 0.1, ~1e-4, 2^24, 1e10
    ties = (np.arange(100000, 1000001, 977, dtype=np.float64) + 0.5)[:, None] * (10.0 ** np.arange(-12, 9))[None, :]
    vals = np.concatenate([bits.view(np.float32), special.view(np.float32), ties.astype(np.float32).ravel(),
                           rng.normal(0, 3, 20000).astype(np.float32), (10.0 ** rng.uniform(-44, 38, 5000)).astype(np.float32)])
    for ncol in (1, 3, 17):
        n = (len(vals) // ncol) * ncol
        rows = vals[:n].reshape(-1, ncol)
        assert E.format_rows(rows) == libc_text(rows), ncol
    assert E.format_rows(np.zeros((0, 4), np.float32)) == b""


@pytest.mark.parametrize("d,n,nsamp,stride", [(16, 300, 12, 1), (2, 7, 5, 1), (8, 1000, 9, 2), (16, 20000, 3, 1)])
def test_samples_text_is_the_text_of_the_rows(d, n, nsamp, stride):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng = M.Engine(d, n, pl=0.9)
    eng.set_option(E.OPT_SAMPLE_STRIDE, stride)
    eng.run(nsamp, 60, O.default_pinit(d, n), vl)
    rows = eng.samples
    kept = (nsamp + stride - 1) // stride
    assert rows.shape == (kept * n, d + 1)
    assert eng.samples_text(0, kept) == libc_text(rows)
    assert eng.samples_text(1, kept - 2) == libc_text(rows[n:(kept - 1) * n])  # a range of steps
    assert eng.samples_text(kept, 0) == b""
    with pytest.raises(M.McxError):
        eng.samples_text(0, kept + 1)


@pytest.mark.parametrize("d,n,nburn,nsamp,pl,block,stride", [
    (16, 512, 120, 103, 0.85, 10, 1),    # one-launch kernel with Murray steps between its launches, ragged last block
    (16, 20000, 55, 45, 0.9, 6, 3),      # hot-path fused kernel, thinned, more blocks than ring slots
    (6, 100, 30, 25, 0.8, 1, 1),         # generic kernel, one step per block
    (2, 5, 0, 7, 1.0, 100, 1),           # block longer than the run
])
def test_text_sink_streams_the_text_of_the_rows(d, n, nburn, nsamp, pl, block, stride):
    """mcx_set_text_sink: the blocks of mcx_set_sink, each as the text of its rows; against the oracle's own sample store
    printed by the C library"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    p = O.default_pinit(d, n)
    vo, keep_o = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=pl, threads=8)
    eo.set_record(samples=True, mask=False, stride=stride)
    eo.run(nsamp, nburn, p, vo)
    vg, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng = M.Engine(d, n, pl=pl)
    eng.set_option(E.OPT_SAMPLE_STRIDE, stride)
    got, calls = [], []

    def sink(first, nsteps, text):
        calls.append((first, nsteps))
        got.append(bytes(text))
        return 0
    eng.set_text_sink(sink, block)
    eng.run(nsamp, nburn, p, vg)
    kb = (block + stride - 1) // stride
    nkeep = (nsamp + stride - 1) // stride
    assert calls == [(f, min(kb, nkeep - f)) for f in range(0, nkeep, kb)]
    assert b"".join(got) == libc_text(eo.samples)
    gl, gp = eng.maxlike()  # the running maximum is kept on the device as with the row sink
    ll = eo.samples[:, d]
    i = int(np.argmax(np.where(ll > -np.inf, ll, -np.inf)))
    assert gl == ll[i] and np.array_equal(gp.view(np.uint32), eo.samples[i, :d].view(np.uint32))
    # a row sink afterwards replaces the text sink
    rows = []
    eng.set_sink(lambda f, ns, r: rows.append(r.copy()) and 0, block)
    eng.run(nsamp, nburn, p, vg)
    assert np.concatenate(rows).shape == eo.samples.shape
    eo.close()


def test_row_sink_can_ask_for_its_blocks_text():
    """MCX_OPT_SINK_TEXT: inside a row sink's callback mcx_sink_text gives the characters of exactly those rows; outside a
    callback (or without the option) it is an error"""
    import mcpar_amd as M
    from mcpar_amd import engine as E
    d, n, nburn, nsamp, block = 16, 777, 60, 23, 5
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng = M.Engine(d, n, pl=0.9)
    eng.set_option(E.OPT_SINK_TEXT, 1)
    seen = []

    def sink(first, nsteps, rows):
        seen.append((rows.copy(), eng.sink_text()))
        return 0
    eng.set_sink(sink, block)
    eng.run(nsamp, nburn, O.default_pinit(d, n), vl)
    assert len(seen) == 5
    for rows, text in seen:
        assert text == libc_text(rows)
    with pytest.raises(M.McxError):
        eng.sink_text()
    eng.set_option(E.OPT_SINK_TEXT, 0)
    errors = []

    def sink2(first, nsteps, rows):
        try:
            eng.sink_text()
        except M.McxError:
            errors.append(first)
        return 0
    eng.set_sink(sink2, block)
    eng.run(nsamp, nburn, O.default_pinit(d, n), vl)
    assert len(errors) == 5
