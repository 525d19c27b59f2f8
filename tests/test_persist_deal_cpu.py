"""Host logic of the one-launch small-n kernel (mcx_k_persist.hip): the table that says which generator wavefront makes
which of a phase's random numbers must cover every step of every (owner, block) set exactly once -- as two-step items or
single steps --, every acceptance item and the 1/pwgt item once, give owners (and, once they record, recorders) nothing,
and load the SIMD that carries an owner less than the others."""
import ctypes as C

import numpy as np
import pytest

END = 0xffffffff
WORDS = 3 * 16 * 24


def deal(lpc2, bpl, own):
    import mcpar_amd as M
    lib = M.load()
    rec, k = C.c_int(), C.c_int()
    tab = np.zeros(WORDS, np.uint32)
    rc = lib.mcx_debug_persist_deal(lpc2, bpl, own, C.byref(rec), C.byref(k), tab.ctypes.data_as(C.POINTER(C.c_uint32)), tab.size)
    assert rc == 0
    return rec.value, k.value, tab.reshape(3, 16, 24)


def lists(tab, t):
    out = []
    for w in range(16):
        lst = [int(v) for v in tab[t, w]]
        n = lst.index(END) if END in lst else len(lst)
        assert all(v == END for v in lst[n:])
        out.append(lst[:n])
    return out


@pytest.mark.parametrize("own", [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("lpc2,bpl", [(1, 1), (2, 1), (4, 1), (8, 1), (1, 2), (2, 2), (4, 2)])
def test_every_step_is_generated_exactly_once(lpc2, bpl, own):
    rec, k, tab = deal(lpc2, bpl, own)
    assert 2 <= k <= 32 and k % 2 == 0
    for t in range(3):
        steps, acc, winv = [], [], 0
        for w, lst in enumerate(lists(tab, t)):
            if w < own or (t > 0 and rec and w < 2 * own):
                assert not lst, (t, w)  # owners never generate; recorders only before the first phase
            for v in lst:
                kind, ob = v >> 14, v & 15
                if kind == 0:
                    steps += [(ob, 2 * ((v >> 4) & 0x3ff)), (ob, 2 * ((v >> 4) & 0x3ff) + 1)]
                elif kind == 3:
                    steps.append((ob, (v >> 4) & 0x3ff))
                elif kind == 1:
                    acc.append(ob)
                else:
                    winv += 1
        assert sorted(steps) == [(ob, s) for ob in range(own * bpl) for s in range(k)], t
        assert sorted(acc) == list(range(own)) and winv == 1, t


def test_the_owners_simd_gets_less():
    rec, k, tab = deal(2, 2, 1)  # 8192 x 16-D with two blocks per lane: owner on SIMD 0, recorder on SIMD 1
    per_simd = [0, 0, 0, 0]
    for w, lst in enumerate(lists(tab, 2)):
        per_simd[w % 4] += sum(2 if v >> 14 == 0 else (1 if v >> 14 == 3 else 0) for v in lst)
    assert per_simd[0] < per_simd[1] <= max(per_simd[2], per_simd[3]), per_simd
    assert sum(per_simd) == k * 2


@pytest.mark.parametrize("lpc2,bpl,own,want_rec,want_k", [
    (2, 1, 1, 1, 32),   # C2: one set of chains per workgroup -> recorders, 32-step phases
    (4, 1, 1, 1, 32),   # 4096 x 16-D
    (4, 1, 2, 0, 32),   # 8192 x 16-D, the strong-scaled shape: two owners, no recorders
    (4, 1, 3, 0, 20),   # 12288 x 16-D
    (2, 2, 2, 0, 16),   # 16384 x 16-D with two blocks per lane
    (2, 2, 3, 0, 10),   # 24576 x 16-D: what the LDS double buffers hold
])
def test_configuration_table(lpc2, bpl, own, want_rec, want_k):
    """the measured configuration rules of the one-launch kernel (EXPERIMENTS.md) as the library applies them"""
    rec, k, tab = deal(lpc2, bpl, own)
    assert (rec, k) == (want_rec, want_k)
