"""MCX_VL_SOURCE, the part that needs no GPU: a user's likelihood given as HIP source (include/mcx.h) builds into the
engine's step kernels against the headers embedded in libmcx.so -- hiprtc cross-compiles for gfx950 like hipcc does --
for every lanes-per-chain the engine has, and a text that does not compile comes back as MCX_ERR_VLFUNC with the
compiler's own message.  The plug-in surface this serves is VLFunc (src/vlfunc.hh:9-12, called at src/mcpar.cc:60,160)."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src(name):
    return open(os.path.join(ROOT, "mcpar_amd", "examples", name)).read().encode()


def compile_for(lib, text, d):
    n = C.c_size_t(0)
    rc = lib.mcx_debug_user_source_compile(text, d, C.byref(n))
    return rc, n.value, lib.mcx_last_error().decode("utf-8", "replace")


def test_hiprtc_is_there_and_the_sample_sources_build():
    import mcpar_amd as M
    lib = M.load()
    assert lib.mcx_user_source_available() == 1, lib.mcx_last_error()
    for name in ("user_rosenbrock1_blocks.hip", "user_rosenbrock1_whole.hip", "user_banana.hip"):
        for d in (2, 6, 16, 40):  # 1, 2, 4 lanes per chain (hot-path + generic kernels) and 16 (generic only)
            rc, nbytes, err = compile_for(lib, src(name), d)
            assert rc == 0 and nbytes > 10000, (name, d, err)


def test_the_one_launch_small_n_kernel_builds_around_a_block_form_text():
    import mcpar_amd as M
    lib = M.load()
    n = C.c_size_t(0)
    for d, bpl, rec in ((8, 1, 1), (16, 1, 0), (16, 2, 0), (32, 2, 0), (4, 1, 1)):
        rc = lib.mcx_debug_user_source_compile_small(src("user_rosenbrock1_blocks.hip"), d, bpl, rec, C.byref(n))
        assert rc == 0 and n.value > 10000, (d, bpl, rec, lib.mcx_last_error())
    rc = lib.mcx_debug_user_source_compile_small(src("user_banana.hip"), 8, 1, 1, None)  # whole-vector form
    assert rc == 7 and b"block form only" in lib.mcx_last_error()


def test_a_text_that_does_not_compile_says_why():
    import mcpar_amd as M
    lib = M.load()
    bad = b"__device__ float mcx_user_loglike(const float *x, int d, const float *par) { return no_such_thing(x[0]); }\n"
    rc, _n, err = compile_for(lib, bad, 16)
    assert rc == 7  # MCX_ERR_VLFUNC
    assert "no_such_thing" in err and "mcx_user_likelihood:1" in err  # the user's own line numbers
    rc, _n, err = compile_for(lib, b"// neither form is defined here\n", 16)
    assert rc == 7 and "mcx_user_loglike" in err
    rc, _n, err = compile_for(lib, b"", 0)
    assert rc == 1  # MCX_ERR_INVALID


def test_symbols_are_declared_and_exported():
    import mcpar_amd as M
    lib = M.load()
    hdr = open(os.path.join(ROOT, "include", "mcx.h")).read()
    for name in ("mcx_user_source_available", "mcx_debug_user_source_compile", "mcx_user_kernel_compile"):
        assert name in hdr and hasattr(lib, name)
    assert "MCX_VL_SOURCE = 102" in hdr and M.VL_SOURCE == 102
