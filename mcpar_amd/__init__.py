"""mcpar_amd -- MI355X-native parallel Metropolis-Hastings engine (drop-in for the chain-step hot
path of rplzzz/mcpar).  The product is libmcx.so (hand-written HIP for gfx950 behind the C ABI of
include/mcx.h); this package is the thin ctypes host binding used by bench.py and the tests.

There is no CPU fallback: importing works anywhere, but every compute call needs the built HIP
library and a GPU, and raises McxError otherwise.
"""
from ._lib import McxError, load, lib_path  # noqa: F401
from .engine import (Engine, VL_DEVICE, VL_SOURCE, VL_DUALGAUSS, VL_GAUSSIAN, VL_GAUSSMIX, VL_HOST,  # noqa: F401
                     VL_ROSENBROCK1, VL_ROSENBROCK2, debug_normals, debug_numerics, device_info,
                     make_vlfunc, vlfunc_eval)

__all__ = ["Engine", "McxError", "load", "lib_path", "make_vlfunc", "vlfunc_eval", "device_info",
           "debug_numerics", "debug_normals"]
