"""ctypes binding of the CPU oracle (oracle/libmcx_oracle.so) -- test infrastructure only.

Nothing under mcpar_amd/ imports this module; it is the checker, never the thing measured.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

VL_ROSENBROCK1, VL_ROSENBROCK2, VL_GAUSSIAN, VL_DUALGAUSS, VL_GAUSSMIX, VL_HOST = 1, 2, 3, 4, 5, 100
VL_ROSENBROCK2_FIXED = 6

HOSTFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))
XFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_int)


class VLFunc(C.Structure):
    _fields_ = [("kind", C.c_int), ("d", C.c_int), ("ncomp", C.c_int),
                ("params", C.POINTER(C.c_float)), ("fn", HOSTFN), ("ctx", C.c_void_p)]


def build_oracle():
    so = os.path.join(ORACLE_DIR, "libmcx_oracle.so")
    src = os.path.join(ORACLE_DIR, "mcx_oracle.c")
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libmcx_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    fp = C.POINTER(C.c_float)
    u32p = C.POINTER(C.c_uint32)
    L.mcxo_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.mcxo_logf.restype = C.c_float
    L.mcxo_logf.argtypes = [C.c_float]
    L.mcxo_expf.restype = C.c_float
    L.mcxo_expf.argtypes = [C.c_float]
    L.mcxo_accept_lu.restype = C.c_float
    L.mcxo_accept_lu.argtypes = [C.c_uint32]
    L.mcxo_sincos2pi.argtypes = [C.c_uint32, fp, fp]
    L.mcxo_u24.restype = C.c_float
    L.mcxo_u24.argtypes = [C.c_uint32]
    L.mcxo_uopen.restype = C.c_float
    L.mcxo_uopen.argtypes = [C.c_uint32]
    L.mcxo_normal4.argtypes = [C.c_uint32] * 6 + [fp]
    L.mcxo_cholesky.argtypes = [C.c_int, fp]
    L.mcxo_vlfunc_eval.argtypes = [C.POINTER(VLFunc), C.c_int, fp, fp]
    L.mcxo_create.restype = C.c_void_p
    L.mcxo_create.argtypes = [C.c_int] * 4 + [C.c_float] * 5 + [C.c_int, C.c_uint32]
    L.mcxo_destroy.argtypes = [C.c_void_p]
    L.mcxo_set_exchange.argtypes = [C.c_void_p, XFN, C.c_void_p]
    L.mcxo_set_threads.argtypes = [C.c_void_p, C.c_int]
    L.mcxo_set_scalar_sweep.argtypes = [C.c_int]
    L.mcxo_set_record.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mcxo_set_sample_stride.argtypes = [C.c_void_p, C.c_int]
    L.mcxo_run.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, C.POINTER(VLFunc), fp]
    L.mcxo_run_all.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(fp),
                               C.POINTER(VLFunc), fp]
    for name in ("state", "loglike", "mean", "var", "musigall", "chol", "samples"):
        f = getattr(L, "mcxo_" + name)
        f.restype = fp
        f.argtypes = [C.c_void_p]
    L.mcxo_accept_counts.restype = u32p
    L.mcxo_accept_counts.argtypes = [C.c_void_p]
    L.mcxo_accept_mask.restype = C.POINTER(C.c_uint8)
    L.mcxo_accept_mask.argtypes = [C.c_void_p]
    for name in ("naccept_burn", "naccept_main", "remote_steps", "remote_passes"):
        f = getattr(L, "mcxo_" + name)
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p]
    L.mcxo_nsample_rows.restype = C.c_size_t
    L.mcxo_nsample_rows.argtypes = [C.c_void_p]
    L.mcxo_tuner_trace.argtypes = [C.c_void_p, fp, C.c_int]
    L.mcxo_gen_local.argtypes = [C.c_void_p, C.c_uint32, fp, fp, fp]
    L.mcxo_gen_remote.argtypes = [C.c_void_p, C.c_uint32, fp, fp, fp, fp, fp, fp, C.POINTER(C.c_int)]
    _lib = L
    return L


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_vlfunc(kind, d, params=None, ncomp=0, fn=None):
    """returns (struct, keepalive)"""
    p = None
    if params is not None:
        p = np.ascontiguousarray(params, dtype=np.float32)
    v = VLFunc(kind, d, ncomp, fptr(p) if p is not None else None,
               fn if fn is not None else HOSTFN(), None)
    return v, (p, fn)


def vl_eval(kind, d, x, params=None, ncomp=0):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, d)
    y = np.empty(x.shape[0], dtype=np.float32)
    v, keep = make_vlfunc(kind, d, params, ncomp)
    st = lib().mcxo_vlfunc_eval(C.byref(v), x.shape[0], fptr(x), fptr(y))
    if st:
        raise ValueError("oracle vlfunc status %d" % st)
    return y


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().mcxo_philox4x32_10(c, k, o)
    return [int(v) for v in o]


class Engine:
    """one shard of an oracle job; mirrors MCPar (src/mcpar.hh:32-37)"""

    def __init__(self, np_, nc, nshards=1, shard=0, pl=0.9, armin=0.2, armax=0.5, dfac=0.2,
                 ifac=1.5, sync=10, seed=8675309, threads=1):
        self.np, self.nc, self.nshards, self.shard = np_, nc, nshards, shard
        self.h = lib().mcxo_create(np_, nc, nshards, shard, pl, armin, armax, dfac, ifac, sync, seed)
        if not self.h:
            raise ValueError("mcxo_create failed")
        lib().mcxo_set_threads(self.h, threads)
        self._keep = []
        self.nburn = self.nsamp = 0

    def close(self):
        if self.h:
            lib().mcxo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_record(self, samples=True, mask=True, stride=1):
        lib().mcxo_set_record(self.h, int(samples), int(mask))
        lib().mcxo_set_sample_stride(self.h, int(stride))

    def set_exchange(self, pyfn):
        def tramp(ctx, musigall, slot, shard, nshards):
            arr = np.ctypeslib.as_array(musigall, shape=(nshards * slot,))
            return int(pyfn(arr, slot, shard, nshards) or 0)
        cb = XFN(tramp)
        self._keep.append(cb)
        lib().mcxo_set_exchange(self.h, cb, None)

    def run(self, nsamp, nburn, pinit, vl, incov=None):
        pinit = np.ascontiguousarray(pinit, dtype=np.float32).reshape(-1)
        assert pinit.size == self.np * self.nc
        ic = None if incov is None else np.ascontiguousarray(incov, dtype=np.float32)
        st = lib().mcxo_run(self.h, nsamp, nburn, fptr(pinit), C.byref(vl),
                            fptr(ic) if ic is not None else None)
        self.nburn, self.nsamp = nburn, nsamp
        if st:
            raise RuntimeError("oracle run status %d" % st)

    def _arr(self, name, shape, dtype=np.float32):
        p = getattr(lib(), "mcxo_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=shape).astype(dtype, copy=True)

    @property
    def state(self): return self._arr("state", (self.nc, self.np))
    @property
    def loglike(self): return self._arr("loglike", (self.nc,))
    @property
    def mean(self): return self._arr("mean", (self.nc, self.np))
    @property
    def var(self): return self._arr("var", (self.nc, self.np))
    @property
    def musigall(self): return self._arr("musigall", (self.nshards * self.nc, self.np, 2))
    @property
    def chol(self): return self._arr("chol", (self.np, self.np))
    @property
    def accept_counts(self): return self._arr("accept_counts", (self.nc,), np.uint32)
    @property
    def naccept_burn(self): return int(lib().mcxo_naccept_burn(self.h))
    @property
    def naccept_main(self): return int(lib().mcxo_naccept_main(self.h))
    @property
    def remote_steps(self): return int(lib().mcxo_remote_steps(self.h))
    @property
    def remote_passes(self): return int(lib().mcxo_remote_passes(self.h))

    @property
    def samples(self):
        n = int(lib().mcxo_nsample_rows(self.h))
        if n == 0:
            return np.zeros((0, self.np + 1), np.float32)
        return self._arr("samples", (n, self.np + 1))

    @property
    def accept_mask(self):
        p = lib().mcxo_accept_mask(self.h)
        return np.ctypeslib.as_array(p, shape=(self.nburn + self.nsamp, self.nc)).copy()

    @property
    def tuner_trace(self):
        buf = np.zeros(256, np.float32)
        n = lib().mcxo_tuner_trace(self.h, fptr(buf), 256)
        return buf[:min(n, 256)].copy()

    def gen_local(self, t, pvals):
        pvals = np.ascontiguousarray(pvals, np.float32)
        pt = np.empty_like(pvals)
        cf = np.empty(self.nc, np.float32)
        lib().mcxo_gen_local(self.h, t, fptr(pvals), fptr(pt), fptr(cf))
        return pt, cf

    def gen_remote(self, t, pvals, musigall):
        pvals = np.ascontiguousarray(pvals, np.float32)
        ms = np.ascontiguousarray(musigall, np.float32)
        pt = np.empty_like(pvals)
        mt = np.empty_like(pvals)
        sg = np.empty_like(pvals)
        cf = np.ones(self.nc, np.float32)
        npass = C.c_int(0)
        lib().mcxo_gen_remote(self.h, t, fptr(pvals), fptr(ms), fptr(pt), fptr(cf), fptr(mt),
                              fptr(sg), C.byref(npass))
        return pt, cf, mt, sg, npass.value


def run_all(engines, nsamp, nburn, pinits, vl, incov=None):
    n = len(engines)
    hs = (C.c_void_p * n)(*[e.h for e in engines])
    ps = [np.ascontiguousarray(p, np.float32).reshape(-1) for p in pinits]
    pp = (C.POINTER(C.c_float) * n)(*[fptr(p) for p in ps])
    ic = None if incov is None else np.ascontiguousarray(incov, np.float32)
    st = lib().mcxo_run_all(hs, n, nsamp, nburn, pp, C.byref(vl), fptr(ic) if ic is not None else None)
    for e in engines:
        e.nburn, e.nsamp = nburn, nsamp
    if st:
        raise RuntimeError("oracle run_all status %d" % st)


def default_pinit(d, n, g0=0):
    """pinit[g][i] = 0.5 sin(0.37 (g d + i)), global chain id g"""
    g = np.arange(g0, g0 + n, dtype=np.float64)[:, None]
    i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)
