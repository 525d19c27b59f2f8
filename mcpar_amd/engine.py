"""Host-side mirror of the reference's MCPar interface (src/mcpar.hh:32-42) over the C ABI."""
import ctypes as C

import numpy as np

from ._lib import (HOSTFN, K_NAMES, OUTFN, SINKFN, TEXTSINKFN, XCHGFN, Counters, PlanItem, Profile, VLFunc, check, load)

VL_ROSENBROCK1, VL_ROSENBROCK2, VL_GAUSSIAN, VL_DUALGAUSS, VL_GAUSSMIX, VL_HOST = 1, 2, 3, 4, 5, 100
VL_DEVICE = 101
VL_SOURCE = 102
VL_ROSENBROCK2_FIXED = 6
OPT_SAMPLES, OPT_ACCEPT_MASK, OPT_FUSE, OPT_MAX_SEGMENT, OPT_PROFILE, OPT_STREAM, OPT_EAGER_EXCHANGE = 1, 2, 3, 4, 5, 6, 7
OPT_SAMPLE_STRIDE = 8
OPT_SPLIT_RNG = 9
OPT_PERSIST = 10
OPT_MEET_TIMEOUT_MS = 11
OPT_DEBUG_MEET = 12
OPT_CULL = 13
OPT_BLOCKS_PER_LANE = 14
OPT_ASYNC_TAIL = 15
OPT_SINK_TEXT = 16
OPT_MEET_UNDER_GATHER = 17
OPT_MURRAY_OVERLAP = 18
OPT_ASYNC_RUN = 19
OPT_REFERENCE_CALLS = 20
OPT_SELF_REPORT = 21
XCHG_BEGIN, XCHG_WAIT = 0, 1


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def format_rows(rows):
    """rows [nrows, ncol] float32 -> the bytes MCout::output prints for them (src/mcout.cc:41-45), formatted on the GPU"""
    rows = np.ascontiguousarray(rows, np.float32)
    nrows, ncol = rows.shape
    nb = C.c_size_t(0)
    check(load().mcx_format_rows(_fp(rows), nrows, ncol, None, 0, C.byref(nb)))
    buf = C.create_string_buffer(max(nb.value, 1))
    check(load().mcx_format_rows(_fp(rows), nrows, ncol, buf, nb.value, C.byref(nb)))
    return buf.raw[:nb.value]


def user_source_available():
    return bool(load().mcx_user_source_available())


def compile_user_kernel(source, symbol):
    """HIP source of a whole kernel with the MCX_VL_DEVICE contract -> its hipFunction_t (int), through hiprtc"""
    fn = C.c_void_p()
    check(load().mcx_user_kernel_compile(source.encode(), symbol.encode(), C.byref(fn)))
    return fn.value


def make_vlfunc(kind, d, params=None, ncomp=0, host_fn=None, device_fn=None, source=None):
    """Build an mcx_vlfunc.  host_fn(x[npset, d]) -> y[npset] wraps a user VLFunc (src/vlfunc.hh:9-12);
    device_fn is a hipFunction_t (int) of a user kernel f(int npset, const float *x, float *y);
    source (VL_SOURCE) is HIP text of the user's device functions, params their `par`.
    Returns (struct, keepalive)."""
    p = None if params is None else np.ascontiguousarray(params, dtype=np.float32)
    if source is not None:
        txt = C.create_string_buffer(source.encode() if isinstance(source, str) else bytes(source))
        v = VLFunc(kind, d, 0 if p is None else p.size, _fp(p) if p is not None else None, HOSTFN(), C.cast(txt, C.c_void_p))
        return v, (p, txt)
    cb = HOSTFN()
    if host_fn is not None:
        def tramp(ctx, npset, x, y):
            xa = np.ctypeslib.as_array(x, shape=(npset, d))
            ya = np.ctypeslib.as_array(y, shape=(npset,))
            ya[:] = np.asarray(host_fn(xa), dtype=np.float32)
            return 0
        cb = HOSTFN(tramp)
    v = VLFunc(kind, d, ncomp, _fp(p) if p is not None else None, cb,
               C.c_void_p(device_fn) if device_fn is not None else None)
    return v, (p, cb)


def vlfunc_eval(kind, d, x, params=None, ncomp=0):
    """VLFunc::operator()(npset, x, y) on the GPU"""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, d)
    y = np.empty(x.shape[0], dtype=np.float32)
    v, keep = make_vlfunc(kind, d, params, ncomp)
    check(load().mcx_vlfunc_eval(C.byref(v), x.shape[0], _fp(x), _fp(y)))
    return y


PLAN_NAMES = {1: "burn_segment", 2: "tuner", 3: "init_moments", 4: "output", 5: "publish", 6: "gather_begin",
              7: "gather_wait", 8: "remote_step", 9: "main_segment", 10: "sink"}


def plan(nsamp, nburn, sync=10, pl=0.9, seed=8675309, tbase=0, nshards=1, eager=0, fused=1, max_segment=256,
         has_output_hook=0, sink_block=0):
    """the launch schedule mcx_run executes for these settings (host logic only, needs no GPU)"""
    n = C.c_int(0)
    check(load().mcx_plan(nsamp, nburn, sync, pl, seed, tbase, nshards, eager, fused, max_segment, has_output_hook,
                          sink_block, None, 0, C.byref(n)))
    items = (PlanItem * max(1, n.value))()
    check(load().mcx_plan(nsamp, nburn, sync, pl, seed, tbase, nshards, eager, fused, max_segment, has_output_hook,
                          sink_block, items, n.value, C.byref(n)))
    return [(PLAN_NAMES[it.kind], it.first, it.nsteps, it.aux) for it in items[:n.value]]


def device_count():
    n = C.c_int(0)
    check(load().mcx_device_count(C.byref(n)))
    return n.value


def rccl_available():
    return bool(load().mcx_rccl_available())


def rccl_unique_id():
    """ncclGetUniqueId: call on one rank, ship the 128 bytes to every rank, pass them to Engine.rccl_init"""
    buf = C.create_string_buffer(128)
    check(load().mcx_rccl_unique_id(buf))
    return buf.raw


def device_info():
    name = C.create_string_buffer(256)
    cu = C.c_int(0)
    mem = C.c_size_t(0)
    check(load().mcx_device_info(name, 256, C.byref(cu), C.byref(mem)))
    return name.value.decode(), cu.value, mem.value


def debug_numerics(what, words):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty_like(w)
    u32p = C.POINTER(C.c_uint32)
    check(load().mcx_debug_numerics(what, w.size, w.ctypes.data_as(u32p), out.ctypes.data_as(u32p)))
    return out


def debug_murray_screen(x, musig, own0=0, sums=True):
    """the per-pair screen of the Murray sweeps alone (mcx_screen.hpp): masks[(N + 63) // 64, (n + 127) // 128] uint64 for
    chains x[n, d] (groups of 128, in this order) against Gaussians musig[N, d, 2] = (mu, sig2)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    musig = np.ascontiguousarray(musig, dtype=np.float32)
    n, d = x.shape
    N = musig.shape[0]
    masks = np.zeros(((N + 63) // 64, (n + 127) // 128), np.uint64)
    check(load().mcx_debug_murray_screen(d, n, N, _fp(x), _fp(musig), int(own0), int(bool(sums)),
                                         masks.ctypes.data_as(C.POINTER(C.c_uint64))))
    return masks


def debug_normals(seed, stream, t, g0, a, q, n):
    out = np.empty((n, 4), np.float32)
    check(load().mcx_debug_normals(seed, stream, t, g0, a, q, n, _fp(out)))
    return out


class Engine:
    """MCPar(np, nc, mpisiz, mpirank, pl, armin, armax, dfac, ifac, sync) -- src/mcpar.hh:32-33 --
    with shards in place of MPI ranks."""

    def __init__(self, np_, nc, nshards=1, shard=0, pl=0.9, armin=0.2, armax=0.5, dfac=0.2,
                 ifac=1.5, sync=10, seed=8675309):
        self.np, self.nc, self.nshards, self.shard = np_, nc, nshards, shard
        self.h = C.c_void_p()
        self._keep = []
        check(load().mcx_create(C.byref(self.h), np_, nc, nshards, shard, pl, armin, armax, dfac,
                                ifac, sync, seed))
        self.nburn = self.nsamp = 0

    def close(self):
        if self.h:
            load().mcx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, value):
        check(load().mcx_set_option(self.h, opt, int(value)))

    def set_exchange(self, pyfn):
        """pyfn(phase, musigall_dev_ptr, slot_floats, shard, nshards, stream_ptr) -> 0 on success"""
        def tramp(ctx, phase, ptr, slot, shard, nshards, stream):
            try:
                return int(pyfn(phase, ptr, slot, shard, nshards, stream) or 0)
            except Exception:  # never unwind through C
                import traceback
                traceback.print_exc()
                return 1
        cb = XCHGFN(tramp)
        self._keep.append(cb)
        check(load().mcx_set_exchange(self.h, cb, None))

    def rccl_init(self, unique_id):
        """install the library's own exchange: in-place ncclAllGather over the nshards engines (collective)"""
        assert len(unique_id) == 128
        self._keep.append(unique_id)
        check(load().mcx_exchange_rccl_init(self.h, C.c_char_p(unique_id)))

    def rccl_init_raw(self, ptr):
        check(load().mcx_exchange_rccl_init(self.h, ptr))

    def rccl_info(self):
        """(ncclCommCount, ncclCommUserRank) of the installed RCCL exchange"""
        nr, rk = C.c_int(0), C.c_int(0)
        check(load().mcx_exchange_rccl_info(self.h, C.byref(nr), C.byref(rk)))
        return nr.value, rk.value

    def rccl_destroy(self):
        check(load().mcx_exchange_rccl_destroy(self.h))

    def debug_exchange(self):
        check(load().mcx_debug_exchange(self.h))

    def synchronize(self):
        """wait for what the last run() left in flight (MCX_OPT_ASYNC_TAIL: a sharded run's last all-gather)"""
        check(load().mcx_synchronize(self.h))

    def exchange_self_check(self):
        """every shard fills its slot with shard + 1, one exchange, then slot r must be full of r + 1 on every
        shard: True / False.  Collective over the shards (each calls it)."""
        check(load().mcx_debug_fill_slot(self.h, float(self.shard + 1)))
        self.debug_exchange()
        ms = self.musigall.reshape(self.nshards, -1)
        ok = bool(all(np.all(ms[r] == np.float32(r + 1)) for r in range(self.nshards)))
        check(load().mcx_debug_fill_slot(self.h, 0.0))  # leave the slots as a fresh engine has them
        self.debug_exchange()
        return ok

    def set_output_hook(self, pyfn):
        def tramp(ctx, steps_done):
            try:
                return int(pyfn(steps_done) or 0)
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = OUTFN(tramp)
        self._keep.append(cb)
        check(load().mcx_set_output_hook(self.h, cb, None))

    def set_sink(self, pyfn, block_steps):
        """pyfn(first_step, nsteps, rows[nsteps*nc, np+1]) -> 0; rows is a view of pinned memory valid during the call.
        pyfn = None removes the sink."""
        if pyfn is None:
            check(load().mcx_set_sink(self.h, SINKFN(), None, 0))
            return
        nc, ncol = self.nc, self.np + 1

        def tramp(ctx, first, nsteps, rows):
            try:
                view = np.ctypeslib.as_array(rows, shape=(nsteps * nc, ncol))
                return int(pyfn(first, nsteps, view) or 0)
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = SINKFN(tramp)
        self._keep.append(cb)
        check(load().mcx_set_sink(self.h, cb, None, int(block_steps)))

    def sink_text(self):
        """inside a row sink's callback of a run with OPT_SINK_TEXT: the block's rows as text (bytes)"""
        ptr, nb = C.c_void_p(0), C.c_size_t(0)
        check(load().mcx_sink_text(self.h, C.byref(ptr), C.byref(nb)))
        return C.string_at(ptr.value, nb.value) if nb.value else b""

    def set_text_sink(self, pyfn, block_steps):
        """pyfn(first_step, nsteps, text: bytes-like memoryview) -> 0: every block as the text MCout::output prints for it
        (src/mcout.cc:41-45), formatted on the device.  pyfn = None removes the sink."""
        if pyfn is None:
            check(load().mcx_set_text_sink(self.h, TEXTSINKFN(), None, 0))
            return

        def tramp(ctx, first, nsteps, text, nbytes):
            try:
                view = (C.c_char * nbytes).from_address(text) if nbytes else b""
                return int(pyfn(first, nsteps, memoryview(view)) or 0)
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = TEXTSINKFN(tramp)
        self._keep.append(cb)
        check(load().mcx_set_text_sink(self.h, cb, None, int(block_steps)))

    def stage_pinit(self, pinit):
        """put the initial chain state in HBM ahead of time; run(..., pinit=None, ...) starts from it"""
        pinit = np.ascontiguousarray(pinit, dtype=np.float32).reshape(-1)
        if pinit.size != self.np * self.nc:
            raise ValueError("pinit must have nc*np elements")
        check(load().mcx_stage_pinit(self.h, _fp(pinit)))

    def run(self, nsamp, nburn, pinit, vl, incov=None):
        """MCPar::run(nsamp, nburn, pinit, L, outsamples, incov) -- src/mcpar.hh:36-37"""
        if pinit is not None:
            pinit = np.ascontiguousarray(pinit, dtype=np.float32).reshape(-1)
            if pinit.size != self.np * self.nc:
                raise ValueError("pinit must have nc*np elements")
        ic = None if incov is None else np.ascontiguousarray(incov, dtype=np.float32)
        check(load().mcx_run(self.h, nsamp, nburn, _fp(pinit) if pinit is not None else None, C.byref(vl),
                             _fp(ic) if ic is not None else None))
        self.nburn, self.nsamp = nburn, nsamp

    def gen_local(self, t, pvals):
        pv = np.ascontiguousarray(pvals, np.float32)
        pt = np.empty_like(pv)
        cf = np.empty(self.nc, np.float32)
        check(load().mcx_gen_local(self.h, t, _fp(pv), _fp(pt), _fp(cf)))
        return pt, cf

    def gen_remote(self, t, pvals, musigall):
        pv = np.ascontiguousarray(pvals, np.float32)
        ms = np.ascontiguousarray(musigall, np.float32)
        pt, mt, sg = np.empty_like(pv), np.empty_like(pv), np.empty_like(pv)
        cf = np.empty(self.nc, np.float32)
        npass = C.c_int(0)
        check(load().mcx_gen_remote(self.h, t, _fp(pv), _fp(ms), _fp(pt), _fp(cf), _fp(mt), _fp(sg),
                                    C.byref(npass)))
        return pt, cf, mt, sg, npass.value

    def covar_setup(self, incov=None):
        out = np.empty((self.np, self.np), np.float32)
        ic = None if incov is None else np.ascontiguousarray(incov, np.float32)
        check(load().mcx_covar_setup(self.h, _fp(ic) if ic is not None else None, _fp(out)))
        return out

    @property
    def state(self): return self._getf("mcx_get_state", (self.nc, self.np))
    @property
    def loglike(self): return self._getf("mcx_get_loglike", (self.nc,))
    @property
    def mean(self): return self._getf("mcx_get_mean", (self.nc, self.np))
    @property
    def var(self): return self._getf("mcx_get_var", (self.nc, self.np))
    @property
    def musigall(self): return self._getf("mcx_get_musigall", (self.nshards * self.nc, self.np, 2))
    @property
    def chol(self): return self._getf("mcx_get_chol", (self.np, self.np))

    def _getf(self, name, shape):
        out = np.empty(shape, np.float32)
        check(getattr(load(), name)(self.h, _fp(out)))
        return out

    @property
    def accept_counts(self):
        out = np.empty(self.nc, np.uint32)
        check(load().mcx_get_accept_counts(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    @property
    def accept_mask(self):
        out = np.empty((self.nburn + self.nsamp, self.nc), np.uint8)
        check(load().mcx_get_accept_mask(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    @property
    def counters(self):
        c = Counters()
        check(load().mcx_get_counters(self.h, C.byref(c)))
        return {n: int(getattr(c, n)) for n, _ in Counters._fields_}

    @property
    def tuner_trace(self):
        buf = np.zeros(256, np.float32)
        n = C.c_int(0)
        check(load().mcx_get_tuner_trace(self.h, _fp(buf), 256, C.byref(n)))
        return buf[:min(n.value, 256)].copy()

    @property
    def samples(self):
        """MCout rows (np+1 columns), step-major then chain (src/mcout.cc:129-145)"""
        ns = C.c_int(0)
        check(load().mcx_samples_steps(self.h, C.byref(ns)))
        out = np.empty((ns.value * self.nc, self.np + 1), np.float32)
        if ns.value:
            check(load().mcx_samples_copy(self.h, 0, ns.value, _fp(out)))
        return out

    def samples_into(self, out, first_step=0, nsteps=None):
        """copy sample rows into a caller-owned float32 array of (nsteps*nc, np+1) elements"""
        if nsteps is None:
            ns = C.c_int(0)
            check(load().mcx_samples_steps(self.h, C.byref(ns)))
            nsteps = ns.value - first_step
        if out.dtype != np.float32 or not out.flags.c_contiguous or out.size < nsteps * self.nc * (self.np + 1):
            raise ValueError("out must be a C-contiguous float32 array of nsteps*nc*(np+1) elements")
        if nsteps:
            check(load().mcx_samples_copy(self.h, first_step, nsteps, _fp(out)))
        return nsteps

    def samples_range(self, first_step, nsteps):
        out = np.empty((nsteps * self.nc, self.np + 1), np.float32)
        if nsteps:
            check(load().mcx_samples_copy(self.h, first_step, nsteps, _fp(out)))
        return out

    def samples_text(self, first_step, nsteps):
        """the rows of samples_range(first_step, nsteps) as the bytes MCout::output prints for them (src/mcout.cc:41-45),
        formatted on the device"""
        nb = C.c_size_t(0)
        check(load().mcx_samples_text(self.h, first_step, nsteps, None, 0, C.byref(nb)))
        buf = C.create_string_buffer(max(nb.value, 1))
        check(load().mcx_samples_text(self.h, first_step, nsteps, buf, nb.value, C.byref(nb)))
        return buf.raw[:nb.value]

    def samples_text_into(self, first_step, nsteps, buf):
        """the same into a caller's uint8 array (None: the size only); returns the number of bytes"""
        nb = C.c_size_t(0)
        if buf is None:
            check(load().mcx_samples_text(self.h, first_step, nsteps, None, 0, C.byref(nb)))
        else:
            check(load().mcx_samples_text(self.h, first_step, nsteps, buf.ctypes.data_as(C.c_char_p), buf.size, C.byref(nb)))
        return nb.value

    def maxlike(self):
        lm = C.c_float(0)
        p = np.empty(self.np, np.float32)
        check(load().mcx_samples_maxlike(self.h, C.byref(lm), _fp(p)))
        return lm.value, p

    @property
    def profile(self):
        p = Profile()
        check(load().mcx_get_profile(self.h, C.byref(p)))
        return {K_NAMES[i]: dict(ms=p.ms[i], launches=int(p.launches[i]), chain_steps=int(p.chain_steps[i]))
                for i in range(len(K_NAMES))}
