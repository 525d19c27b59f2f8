"""The per-pair screen of the Murray sweeps on its own (mcx_screen.hpp through mcx_debug_murray_screen): whatever the
inputs, a (group of 128 chains, Gaussian) row may be marked skippable only if NO chain of the group has a float arg at or
below its bound -- 176 for the sum sweeps, min(176, the chain's arg against its own Gaussian) for the min-arg sweep.  The
args here are accumulated as the sweep accumulates them (float32: sub, mul, fma in ascending dimension; the fma through
float64, whose product of two float32 is exact).  Also: on ordinary states the screen must find most of the rows that
really are skippable -- a screen that marks nothing passes the first test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ZERO_ARG = np.float32(176.0)


def sweep_args(x, musig):
    """arg[j, i] with the sweep's operations and order"""
    mu = musig[:, :, 0].astype(np.float32)
    with np.errstate(all="ignore"):
        w = (np.float32(1.0) / musig[:, :, 1].astype(np.float32)).astype(np.float32)
        arg = np.zeros((x.shape[0], mu.shape[0]), np.float32)
        for k in range(x.shape[1]):
            xm = (mu[None, :, k] - x[:, None, k]).astype(np.float32)
            t = (xm * xm).astype(np.float32)
            arg = (t.astype(np.float64) * w[None, :, k].astype(np.float64) + arg.astype(np.float64)).astype(np.float32)
    return arg


def check_masks(x, musig, own0, sums, masks):
    n, N = x.shape[0], musig.shape[0]
    arg = sweep_args(x, musig)
    if sums:
        lim = np.full(n, ZERO_ARG)
    else:
        a0 = arg[np.arange(n), own0 + np.arange(n)]
        lim = np.where(a0 < ZERO_ARG, a0, ZERO_ARG).astype(np.float32)
    with np.errstate(invalid="ignore"):
        matters = ~(arg > lim[:, None])  # (an arg or a bound that is no number: the row matters)
    ng = (n + 127) // 128
    bits = ((masks[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).astype(bool)  # [word][group][bit]
    bits = bits.transpose(1, 0, 2).reshape(ng, -1)                                                                 # [group][Gaussian]
    assert not bits[:, N:].any(), "bits beyond N"
    bits = bits[:, :N]
    need = np.zeros((ng, N), bool)
    for g in range(ng):
        need[g] = matters[g * 128:(g + 1) * 128].any(axis=0)
    missed = need & ~bits
    assert not missed.any(), "rows skipped that matter: %s" % (np.argwhere(missed)[:5],)
    return need.mean(), bits.mean()


def state(rng, N, d, n, spread, width):
    ms = np.empty((N, d, 2), np.float32)
    ms[:, :, 0] = rng.normal(0.4, spread, (N, d))
    ms[:, :, 1] = rng.uniform(0.3 * width, width, (N, d)) ** 2
    x = (ms[:n, :, 0] + np.sqrt(ms[:n, :, 1]) * rng.standard_normal((n, d))).astype(np.float32)
    return ms, x


@pytest.mark.parametrize("d", [16, 32])
@pytest.mark.parametrize("sums", [True, False])
def test_never_skips_a_row_that_matters_and_finds_most_that_do_not(d, sums):
    from mcpar_amd.engine import debug_murray_screen
    rng = np.random.default_rng(d + int(sums))
    for N, n, spread, width in ((1500, 700, 0.4, 0.1), (777, 777, 0.4, 0.02), (1024, 1024, 1.0, 1.0), (300, 129, 0.05, 0.2),
                                (4096, 256, 3.0, 0.3)):
        ms, x = state(rng, N, d, n, spread, width)
        order = np.argsort(x[:, 0], kind="stable")  # (neighbours in one coordinate: something for a group to share)
        own0 = 0
        xs = x[order] if sums else x                # (the min-arg bound names chain j's own Gaussian: keep the order)
        masks = debug_murray_screen(xs, ms, own0, sums)
        need, kept = check_masks(xs, ms, own0, sums, masks)
        print("d %d %s N %d n %d spread %.2f width %.2f: rows that matter %.4f, rows kept %.4f" % (
            d, "sums" if sums else "min-arg", N, n, spread, width, need, kept))
        assert kept <= need + 0.25 * (1.0 - need) + 0.02, "the screen keeps far more than it has to"


@pytest.mark.parametrize("d", [16, 32])
def test_inputs_that_are_no_ordinary_numbers_exclude_nothing_they_should_not(d):
    from mcpar_amd.engine import debug_murray_screen
    rng = np.random.default_rng(7 * d)
    N = n = 640
    for sums in (True, False):
        ms, x = state(rng, N, d, n, 0.4, 0.1)
        ms[3, :, 1] = 0.0                      # 1 / sig2 = inf
        ms[4, 2, 1] = np.nan
        ms[5, :, 1] = -1.0                     # a negative "variance"
        ms[6, :, 1] = 1e-30                    # an enormous weight
        ms[7, :, 0] = 1e6                      # a Gaussian far from everything
        ms[8, 1, 0] = np.inf
        ms[9, :, 1] = 1e30                     # a Gaussian as wide as the world: arg ~ 0 for everybody
        ms[10, :, 0] = 3e19                    # beyond the norms the screen trusts itself with
        x[0, 0] = np.nan
        x[1, 3] = np.inf
        x[2] = 1e20
        x[130] = ms[20, :, 0]                  # a chain exactly at another Gaussian's centre
        x[131:200] = x[131]                    # identical chains
        x[300] = 50.0
        x[301] = -1e6
        masks = debug_murray_screen(x, ms, 0, sums)
        check_masks(x, ms, 0, sums, masks)


def test_far_from_the_origin_the_cancellation_is_still_bounded():
    """everything sits near 1e4 with widths of 1e-2: c_i and A.B are ~1e12 each and cancel to ~1e1"""
    from mcpar_amd.engine import debug_murray_screen
    rng = np.random.default_rng(11)
    d, N, n = 16, 1024, 512
    ms, x = state(rng, N, d, n, 0.4, 0.01)
    ms[:, :, 0] += np.float32(1e4)
    x = (ms[:n, :, 0] + np.sqrt(ms[:n, :, 1]) * rng.standard_normal((n, d))).astype(np.float32)
    for sums in (True, False):
        masks = debug_murray_screen(x, ms, 0, sums)
        need, kept = check_masks(x, ms, 0, sums, masks)
        print("offset 1e4: rows that matter %.4f, kept %.4f" % (need, kept))


@pytest.mark.parametrize("d", [16, 32])
@pytest.mark.parametrize("sums", [True, False])
def test_adversarial_states_never_lose_a_row_that_matters(d, sums):
    """the exactness argument assumes an error model for the matrix core's accumulation that the hardware is
    not documented to follow.  States built to sit on its weak spots -- where a wrongly skipped row would silently change
    qisum / qimax (src/mcpar.cc:367-395): (a) a centre far from the chains, so the products that must cancel are orders of
    magnitude above the arg that is left; (b) Gaussians placed so that a chain's arg is within 1e-6 .. 1e-2 of its bound,
    on either side; (c) weights and coordinates of very different magnitudes from one dimension to the next."""
    from mcpar_amd.engine import debug_murray_screen
    rng = np.random.default_rng(1000 + d + int(sums))
    # (a) most Gaussians (and with them the screen's centre) around the origin, the chains -- and the few Gaussians near
    #     them -- in a tight cloud far away: |A_j| |B_i| ~ (30 sqrt(d))^2 / width^2 against args of tens
    N, n = 2048, 512
    ms = np.empty((N, d, 2), np.float32)
    ms[:, :, 0] = rng.normal(0.0, 0.3, (N, d))
    ms[:, :, 1] = rng.uniform(0.05, 0.2, (N, d)) ** 2
    ms[:n, :, 0] = np.float32(30.0) + rng.normal(0.0, 0.3, (n, d))
    x = (ms[:n, :, 0] + np.sqrt(ms[:n, :, 1]) * rng.standard_normal((n, d)) * 1.5).astype(np.float32)
    masks = debug_murray_screen(x, ms, 0, sums)
    need, kept = check_masks(x, ms, 0, sums, masks)
    print("far centre: rows that matter %.4f, kept %.4f" % (need, kept))
    # (b) unit weights; Gaussian n + i sits at distance sqrt(L (1 + eps_i)) from chain (i mod n) in a random direction,
    #     L = 176 (sum sweep) or the chain's own arg (min-arg sweep, made ~40 by construction)
    N, n = 1024, 256
    ms = np.empty((N, d, 2), np.float32)
    ms[:, :, 1] = 1.0
    ms[:, :, 0] = rng.normal(0.0, 1.0, (N, d)) * 40.0   # far from everybody unless placed below
    x = rng.normal(0.0, 1.0, (n, d)).astype(np.float32)
    own = x + (rng.standard_normal((n, d)) * np.sqrt(40.0 / d)).astype(np.float32)   # own arg ~ 40
    ms[:n, :, 0] = own
    a_own = ((ms[:n, :, 0].astype(np.float64) - x) ** 2).sum(axis=1)
    eps = np.array([s * 10.0 ** e for e in (-6, -5, -4, -3, -2) for s in (-1.0, 1.0)])
    for i in range(n, N):
        j = (i - n) % n
        L = 176.0 if sums else min(176.0, a_own[j])
        u = rng.standard_normal(d)
        u /= np.linalg.norm(u)
        ms[i, :, 0] = (x[j].astype(np.float64) + np.sqrt(L * (1.0 + eps[(i - n) % eps.size])) * u).astype(np.float32)
    masks = debug_murray_screen(x, ms, 0, sums)
    need, kept = check_masks(x, ms, 0, sums, masks)
    print("at the bounds: rows that matter %.4f, kept %.4f" % (need, kept))
    # (c) per-dimension scales from 1e-3 to 1e3 (coordinates) with weights that undo them (every dimension contributes
    #     alike to arg, none alike to |A| |B|), shuffled so that neighbouring dimensions differ by orders of magnitude
    N, n = 1536, 384
    scale = (10.0 ** rng.permutation(np.linspace(-3, 3, d))).astype(np.float32)
    ms = np.empty((N, d, 2), np.float32)
    ms[:, :, 0] = rng.normal(0.4, 1.0, (N, d)) * scale
    ms[:, :, 1] = (rng.uniform(0.1, 0.3, (N, d)) * scale) ** 2
    x = (ms[:n, :, 0] + np.sqrt(ms[:n, :, 1]) * rng.standard_normal((n, d)) * 2.0).astype(np.float32)
    masks = debug_murray_screen(x, ms, 0, sums)
    need, kept = check_masks(x, ms, 0, sums, masks)
    print("mixed magnitudes: rows that matter %.4f, kept %.4f" % (need, kept))
