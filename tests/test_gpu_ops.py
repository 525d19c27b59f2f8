"""The two public proposal generators (MCPar::genLocal / genRemote, src/mcpar.hh:40-42) and
covar_setup through the C ABI, vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,n", [(2, 5), (8, 1000), (16, 4097), (12, 130), (32, 64), (50, 37), (256, 11)])
def test_gen_local(d, n):
    import mcpar_amd as M
    x = np.random.default_rng(d).normal(size=(n, d)).astype(np.float32)
    eo, eg = O.Engine(d, n), M.Engine(d, n)
    for t in (0, 7, 123456):
        po, co = eo.gen_local(t, x)
        pg, cg = eg.gen_local(t, x)
        assert np.array_equal(pg.view(np.uint32), po.view(np.uint32))
        assert np.all(cg == 1.0) and np.all(co == 1.0)
    # proposals are x + z with unit normals under the identity factor
    z = (pg - x).astype(np.float64)
    if n * d > 5000:
        assert abs(z.mean()) < 0.05 and abs(z.var() - 1) < 0.05


def test_gen_local_full_covariance_statistics():
    import mcpar_amd as M
    d, n = 4, 60000
    cov = np.array([[2, .6, .2, 0], [.6, 1, .3, .1], [.2, .3, 1.5, .4], [0, .1, .4, 1]], np.float32)
    eg = M.Engine(d, n)
    f = eg.covar_setup(cov)
    np.testing.assert_allclose(f @ f.T, cov, rtol=1e-5, atol=1e-5)
    x = np.zeros((n, d), np.float32)
    pg, _ = eg.gen_local(5, x)
    np.testing.assert_allclose(np.cov(pg.T.astype(np.float64)), cov, atol=0.04)
    with pytest.raises(M.McxError):
        eg.covar_setup(-np.eye(d, dtype=np.float32))


# (32, 513, 1) / (32, 1030, 2): two chains per lane -- an odd count, and counts that cross one and two workgroups' worth
# (2 x 256 positions each); (16, 700, 1) / (4, 520, 1): N not a multiple of the 256-row block nor of the LDS stage
@pytest.mark.parametrize("d,n,nshards", [(2, 64, 1), (16, 300, 1), (8, 96, 3), (5, 40, 1), (32, 33, 2), (33, 40, 1), (48, 70, 1), (64, 65, 2),
                                         (65, 30, 1), (80, 24, 2), (32, 513, 1), (32, 1030, 2), (16, 700, 1), (4, 520, 1)])
def test_gen_remote(d, n, nshards):
    import mcpar_amd as M
    rng = np.random.default_rng(10 * d + nshards)
    N = n * nshards
    ms = np.empty((N, d, 2), np.float32)
    ms[:, :, 0] = rng.normal(0, 1.0, (N, d))
    ms[:, :, 1] = rng.uniform(0.05, 0.6, (N, d))
    x = rng.normal(0, 1.0, (n, d)).astype(np.float32)
    shard = nshards - 1
    eo, eg = O.Engine(d, n, nshards=nshards, shard=shard), M.Engine(d, n, nshards=nshards, shard=shard)
    po, co, mo, so, npo = eo.gen_remote(42, x, ms)
    pg, cg, mg, sg, npg = eg.gen_remote(42, x, ms)
    assert npg == npo and npo >= 1
    for a, b, name in ((pg, po, "ptrial"), (cg, co, "cfac"), (mg, mo, "mutrial"), (sg, so, "sigtrial")):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
    assert np.all(np.isfinite(cg)) and np.all(cg >= 0)  # 0 when every Q_i underflows at pvals
