"""MCX_VL_DEVICE: a user likelihood supplied as a GPU kernel from the user's own code object; the
whole step stays on the device (no host round trip)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_user_kernel_likelihood(tmp_path):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    M.load()
    co = str(tmp_path / "user.co")
    subprocess.check_call(["hipcc", "--genco", "--offload-arch=gfx950", "-O2", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "cpp", "user_vlfunc_kernel.hip"), "-o", co])
    hip = C.CDLL("libamdhip64.so")
    mod, fn = C.c_void_p(), C.c_void_p()
    assert hip.hipModuleLoad(C.byref(mod), co.encode()) == 0
    assert hip.hipModuleGetFunction(C.byref(fn), mod, b"user_rosenbrock8") == 0
    d, n, nburn, nsamp = 8, 300, 120, 40
    p = O.default_pinit(d, n)
    vo, _k1 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=0.85)
    eo.run(nsamp, nburn, p, vo)
    vg, _k2 = M.make_vlfunc(M.VL_DEVICE, d, device_fn=fn.value)
    eg = M.Engine(d, n, pl=0.85)
    eg.set_option(E.OPT_ACCEPT_MASK, 1)
    eg.run(nsamp, nburn, p, vg)
    assert np.array_equal(eg.accept_mask, eo.accept_mask)
    for name in ("state", "loglike", "mean", "var", "samples"):
        assert np.array_equal(getattr(eg, name).view(np.uint32), getattr(eo, name).view(np.uint32)), name
    # the functor call itself
    x = np.random.default_rng(1).normal(size=(1000, d)).astype(np.float32)
    y = np.empty(1000, np.float32)
    M._lib.check(M.load().mcx_vlfunc_eval(C.byref(vg), 1000, x.ctypes.data_as(C.POINTER(C.c_float)),
                                          y.ctypes.data_as(C.POINTER(C.c_float))))
    assert np.array_equal(y.view(np.uint32), O.vl_eval(O.VL_ROSENBROCK1, d, x).view(np.uint32))
    with pytest.raises(M.McxError):
        bad, _k3 = M.make_vlfunc(M.VL_DEVICE, d)
        eg.run(5, 5, p, bad)
    hip.hipModuleUnload(mod)
