#!/usr/bin/env python3
"""One-launch small-n kernel, two builds of the library in turn on one box (the package's libmcx.so against build/ab/libmcx_base.so,
made in the build container from the other version of mcx_persist.hpp), several shapes, one process per point.  usage: persist_ab2.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from persist_bpl_probe import one  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for d, n in ((16, 8192), (8, 4096), (16, 16384), (32, 8192), (16, 4096), (8, 16384)):
        os.environ.pop("MCX_LIBMCX", None)
        a = one(d, n, 0)
        os.environ["MCX_LIBMCX"] = os.path.join(ROOT, "build", "ab", "libmcx_base.so")
        b = one(d, n, 0)
        print("d=%d n=%d: new %s | base %s" % (d, n, a, b), flush=True)
