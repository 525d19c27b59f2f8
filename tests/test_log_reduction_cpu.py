"""The range reduction of the MCX logarithm (mcx_numerics.hpp: log_reduce) is written without compare and select; the
oracle (oracle/mcx_oracle.c: mcx_logf) writes the text-book form.  Both must give the same exponent and the same mantissa
bits for every float in [0, inf): checked here for EVERY mantissa at exponents across the range (the algebra does not
depend on the exponent), in numpy -- the device code itself is compared with the oracle in tests/test_gpu_numerics.py."""
import numpy as np
import pytest


def textbook(b):
    e = ((b >> np.uint32(23)) & np.uint32(0xFF)).astype(np.int32) - 126
    m = ((b & np.uint32(0x7FFFFF)) | np.uint32(0x3F000000)).view(np.float32)
    lo = m < np.float32(0.70710678)
    return np.where(lo, e - 1, e), np.where(lo, (m + m) - np.float32(1), m - np.float32(1)).astype(np.float32)


def integer_form(b):
    ix = b - np.uint32(0x3F3504F3)
    return ix.view(np.int32) >> 23, (((ix & np.uint32(0x7FFFFF)) + np.uint32(0x3F3504F3)).view(np.float32) - np.float32(1)).astype(np.float32)


@pytest.mark.parametrize("E", [0, 1, 95, 126, 127, 128, 254])
def test_integer_form_equals_textbook_form(E):
    assert np.float32(0.70710678).view(np.uint32) == 0x3F3504F3
    b = (np.uint32(E) << np.uint32(23)) | np.arange(1 << 23, dtype=np.uint32)
    e1, m1 = textbook(b)
    e2, m2 = integer_form(b)
    assert np.array_equal(e1, e2)
    assert np.array_equal(m1.view(np.uint32), m2.view(np.uint32))
