"""Device arithmetic primitives vs the oracle: bit-exact on dense random and edge inputs."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def oracle_map(fn, words):
    return np.array([fn(int(w)) for w in words])


def test_philox_word_stream():
    import mcpar_amd as M
    w = np.random.default_rng(0).integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32)
    got = M.debug_numerics(6, w)
    exp = np.array([O.philox([int(x), 0, 0, 0], [0, 0])[0] for x in w], np.uint32)
    assert np.array_equal(got, exp)


def test_log_exp_sincos_uniform_bits():
    import mcpar_amd as M
    L = O.lib()
    rng = np.random.default_rng(1)
    # logf on (0,1] values produced exactly like the sampler does, plus arbitrary positive normals
    w = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    u = np.array([L.mcxo_uopen(int(x)) for x in w], np.float32)
    assert np.array_equal(M.debug_numerics(5, w), u.view(np.uint32))
    xs = np.concatenate([u, rng.uniform(1e-30, 1e30, 2000).astype(np.float32),
                         np.float32(2.0) ** np.arange(-120, 120, dtype=np.float32)]).astype(np.float32)
    exp = np.array([L.mcxo_logf(float(x)) for x in xs], np.float32)
    assert np.array_equal(M.debug_numerics(0, xs.view(np.uint32)), exp.view(np.uint32))
    # expf incl. clamps, +-0, NaN, inf
    xe = np.concatenate([rng.uniform(-100, 95, 20000), [0.0, -0.0, 88.72283, 88.7229, -87.33654, -87.3366,
                                                        np.inf, -np.inf, np.nan, 1e-30, -1e-30, 1e30, -1e30, 3e38],
                         # both sides of every clamp: n = -126/-125 and n = 127/128 (n = floor(x log2 e + 1/2))
                         np.float32(-125.5 * np.log(2.0)) + np.arange(-40, 41) * np.float32(2.0 ** -17),
                         np.float32(127.5 * np.log(2.0)) + np.arange(-40, 41) * np.float32(2.0 ** -17)]).astype(np.float32)
    exp = np.array([L.mcxo_expf(float(x)) for x in xe], np.float32)
    got = M.debug_numerics(1, xe.view(np.uint32)).view(np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))
    # sincos on random words and every quadrant boundary
    ww = np.concatenate([w, np.array([0, 1, 0x1fffffff, 0x20000000, 0x20000001, 0x3fffffff, 0x40000000,
                                      0x5fffffff, 0x60000000, 0x7fffffff, 0x80000000, 0x9fffffff, 0xa0000000,
                                      0xbfffffff, 0xc0000000, 0xdfffffff, 0xe0000000, 0xffffffff], np.uint32)])
    s, c = C.c_float(), C.c_float()
    es, ec = [], []
    for x in ww:
        L.mcxo_sincos2pi(int(x), C.byref(s), C.byref(c))
        es.append(s.value)
        ec.append(c.value)
    assert np.array_equal(M.debug_numerics(2, ww), np.array(es, np.float32).view(np.uint32))
    assert np.array_equal(M.debug_numerics(3, ww), np.array(ec, np.float32).view(np.uint32))
    assert np.array_equal(M.debug_numerics(4, ww), np.array([L.mcxo_u24(int(x)) for x in ww], np.float32).view(np.uint32))
    # log of the acceptance draw (local steps test log u < ly' - ly): -inf at u = 0, scalar and packed forms
    wa = np.concatenate([ww, np.array([0x00, 0xff, 0x100, 0x1ff, 0x200, 0xffffff00], np.uint32)])
    elu = np.array([L.mcxo_accept_lu(int(x)) for x in wa], np.float32)
    assert elu[-6] == -np.inf and elu[-5] == -np.inf and np.isfinite(elu[-4])
    assert np.array_equal(M.debug_numerics(14, wa), elu.view(np.uint32))
    assert np.array_equal(M.debug_numerics(15, wa), elu.view(np.uint32))


def test_normals_bit_exact_all_streams():
    import mcpar_amd as M
    L = O.lib()
    z = np.zeros(4, np.float32)
    for stream, t, g0, a, q in ((0, 0, 0, 0, 0), (0, 1234, 65536 * 7, 3, 0), (4, 99, 17, 250, 7)):
        got = M.debug_normals(8675309, stream, t, g0, a, q, 3000)
        exp = np.empty((3000, 4), np.float32)
        for i in range(3000):
            L.mcxo_normal4(8675309, stream, t, g0 + i, a, q, O.fptr(z))
            exp[i] = z
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_sqrt_rn_exhaustive():
    """the kernels' lean sqrt (raw v_sqrt_f32 + residual fix-up, no denormal pre-scaling) equals IEEE
    sqrtf on EVERY float in [2^-96, 2^20) and on +-0; the Box-Muller argument -2 ln(u) lies in
    {-0} U [1.19e-7, 44.4].  (Below 2^-102 the residuals underflow and it is NOT exact: that is what
    hipcc's pre-scaling is for.)"""
    import ctypes as C
    import mcpar_amd as M
    nbad, first = C.c_uint64(0), C.c_uint32(0)
    lo, hi = 0x0f800000, 0x49800000  # 2^-96 .. 2^20
    M._lib.check(M.load().mcx_debug_sqrt_sweep(lo, hi, C.byref(nbad), C.byref(first)))
    assert nbad.value == 0, "first mismatch at bits 0x%08x" % first.value
    z = np.array([0.0, -0.0], np.float32).view(np.uint32)
    assert np.array_equal(M.debug_numerics(7, z), M.debug_numerics(8, z))
    # and against the host's sqrtf on a random sample of the Box-Muller domain
    x = np.random.default_rng(4).uniform(1e-7, 45.0, 200000).astype(np.float32)
    assert np.array_equal(M.debug_numerics(7, x.view(np.uint32)), np.sqrt(x).view(np.uint32))


def test_packed_forms_equal_scalar_forms():
    """the 2-wide (v_pk_*_f32) log / exp / sincos used by the hot-path kernel give the same bits as the
    scalar forms, in either lane, including out-of-range and special inputs"""
    import mcpar_amd as M
    rng = np.random.default_rng(11)
    xe = np.concatenate([rng.uniform(-100, 95, 50000), [0.0, -0.0, 88.72283, 88.7229, -87.33654, -87.3366,
                                                        np.inf, -np.inf, 1e-30, -1e-30]]).astype(np.float32)
    ref = M.debug_numerics(1, xe.view(np.uint32))
    assert np.array_equal(M.debug_numerics(9, xe.view(np.uint32)), ref)
    assert np.array_equal(M.debug_numerics(10, xe.view(np.uint32)), ref)
    nan = np.array([np.nan], np.float32).view(np.uint32)
    assert np.isnan(M.debug_numerics(9, nan).view(np.float32)[0])
    xl = np.concatenate([rng.uniform(1e-30, 1e30, 20000), rng.random(30000) + 1e-9]).astype(np.float32)
    assert np.array_equal(M.debug_numerics(11, xl.view(np.uint32)), M.debug_numerics(0, xl.view(np.uint32)))
    w = rng.integers(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(M.debug_numerics(12, w), M.debug_numerics(13, w))
