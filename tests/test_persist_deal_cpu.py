"""Host logic of the one-launch small-n kernel (mcx_k_persist.hip): the table that says which generator wavefront makes
which of a phase's random numbers must hold every item exactly once, give owners (and, once they record, recorders)
nothing, and load the SIMD that carries an owner less than the others."""
import ctypes as C

import numpy as np
import pytest


def deal(lpc2, bpl, own):
    import mcpar_amd as M
    lib = M.load()
    rec, k = C.c_int(), C.c_int()
    tab = np.zeros(3 * 16 * 12, np.uint32)
    rc = lib.mcx_debug_persist_deal(lpc2, bpl, own, C.byref(rec), C.byref(k), tab.ctypes.data_as(C.POINTER(C.c_uint32)), tab.size)
    assert rc == 0
    return rec.value, k.value, tab.reshape(3, 16, 12)


@pytest.mark.parametrize("own", [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("lpc2,bpl", [(1, 1), (2, 1), (4, 1), (8, 1), (1, 2), (2, 2), (4, 2)])
def test_every_item_is_dealt_exactly_once(lpc2, bpl, own):
    rec, k, tab = deal(lpc2, bpl, own)
    assert 2 <= k <= 32
    want = sorted([gp << 4 | ob for gp in range((k + 1) // 2) for ob in range(own * bpl)] + [1 << 14 | o for o in range(own)] + [2 << 14])
    for t in range(3):
        got = []
        for w in range(16):
            lst = list(tab[t, w])
            n = lst.index(0xffffffff) if 0xffffffff in lst else 12
            assert all(v == 0xffffffff for v in lst[n:])
            if w < own or (t > 0 and rec and w < 2 * own):
                assert n == 0, (t, w)  # owners never generate; recorders only before the first phase
            got += lst[:n]
        assert sorted(got) == want, t


def test_the_owners_simd_gets_less():
    rec, k, tab = deal(2, 2, 1)  # 8192 x 16-D with two blocks per lane: owner on SIMD 0, recorder on SIMD 1
    per_simd = [0, 0, 0, 0]
    for w in range(16):
        per_simd[w % 4] += sum(1 for v in tab[2, w] if v != 0xffffffff and v >> 14 == 0)
    assert per_simd[0] <= per_simd[1] and per_simd[0] < max(per_simd[2], per_simd[3]), per_simd
