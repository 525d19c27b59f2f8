"""ctypes loader for libmcx.so.  Fails loudly when the HIP library has not been built."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))


class McxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mcx status %d: %s" % (code, msg))
        self.code = code


HOSTFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))
XCHGFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
OUTFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
SINKFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float))
TEXTSINKFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t)


class VLFunc(C.Structure):
    _fields_ = [("kind", C.c_int), ("d", C.c_int), ("ncomp", C.c_int),
                ("params", C.POINTER(C.c_float)), ("fn", HOSTFN), ("ctx", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("naccept_burn", "naccept_main", "nsteps_burn", "nsteps_main",
                                          "remote_steps", "remote_passes", "exchanges", "kernel_launches",
                                          "remote_pairs", "remote_pairs_evaluated", "meet_timeouts", "meet_timeouts_total",
                                          "small_n_launches", "small_n_blocks_per_lane", "exchange_waits",
                                          "exchange_wait_ns")]


K_NAMES = ("fused_burn", "fused_main", "propose", "eval", "accept", "remote", "tuner", "misc", "remote_sweep",
           "gen_normals", "run_small", "remote_screen")


class PlanItem(C.Structure):
    _fields_ = [("kind", C.c_int), ("first", C.c_int), ("nsteps", C.c_int), ("aux", C.c_int)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * 12), ("launches", C.c_uint64 * 12), ("chain_steps", C.c_uint64 * 12)]


def lib_path():
    # MCX_LIBMCX: another build of the same library (tools/persist_ab.py times variants of one kernel on one box)
    return os.environ.get("MCX_LIBMCX") or os.path.join(HERE, "libmcx.so")


_lib = None
ABI_VERSION = 4  # include/mcx.h MCX_ABI_VERSION


def load():
    """dlopen libmcx.so and declare every entry point of include/mcx.h"""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise McxError(-1, "%s not found: build it with `make lib` (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback" % p)
    L = C.CDLL(p)
    fp, u32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
    sig = {
        "mcx_vlfunc_eval": [C.POINTER(VLFunc), C.c_int, fp, fp],
        "mcx_create": [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                       C.c_float, C.c_float, C.c_float, C.c_int, C.c_uint32],
        "mcx_destroy": [vp],
        "mcx_run": [vp, C.c_int, C.c_int, fp, C.POINTER(VLFunc), fp],
        "mcx_stage_pinit": [vp, fp],
        "mcx_gen_local": [vp, C.c_uint32, fp, fp, fp],
        "mcx_gen_remote": [vp, C.c_uint32, fp, fp, fp, fp, fp, fp, C.POINTER(C.c_int)],
        "mcx_covar_setup": [vp, fp, fp],
        "mcx_set_exchange": [vp, XCHGFN, vp],
        "mcx_set_output_hook": [vp, OUTFN, vp],
        "mcx_set_sink": [vp, SINKFN, vp, C.c_int],
        "mcx_set_text_sink": [vp, TEXTSINKFN, vp, C.c_int],
        "mcx_sink_text": [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
        "mcx_set_option": [vp, C.c_int, C.c_int64],
        "mcx_get_counters": [vp, C.POINTER(Counters)],
        "mcx_get_state": [vp, fp], "mcx_get_loglike": [vp, fp], "mcx_get_mean": [vp, fp],
        "mcx_get_var": [vp, fp], "mcx_get_musigall": [vp, fp], "mcx_get_chol": [vp, fp],
        "mcx_synchronize": [vp],
        "mcx_get_accept_counts": [vp, u32p],
        "mcx_get_accept_mask": [vp, C.POINTER(C.c_uint8)],
        "mcx_get_tuner_trace": [vp, fp, C.c_int, C.POINTER(C.c_int)],
        "mcx_samples_steps": [vp, C.POINTER(C.c_int)],
        "mcx_samples_copy": [vp, C.c_int, C.c_int, fp],
        "mcx_samples_text": [vp, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)],
        "mcx_format_rows": [fp, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)],
        "mcx_samples_maxlike": [vp, fp, fp],
        "mcx_get_profile": [vp, C.POINTER(Profile)],
        "mcx_copy_to_host": [vp, vp, C.c_size_t, vp],
        "mcx_copy_to_device": [vp, vp, C.c_size_t, vp],
        "mcx_plan": [C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                     C.c_int, C.c_int, C.c_int, C.POINTER(PlanItem), C.c_int, C.POINTER(C.c_int)],
        "mcx_abi_version": [],
        "mcx_device_info": [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)],
        "mcx_set_device": [C.c_int],
        "mcx_device_count": [C.POINTER(C.c_int)],
        "mcx_device_pci_bus_id": [C.c_char_p, C.c_size_t],
        "mcx_rccl_available": [],
        "mcx_user_source_available": [],
        "mcx_debug_user_source_compile": [C.c_char_p, C.c_int, C.POINTER(C.c_size_t)],
        "mcx_debug_user_source_compile_small": [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)],
        "mcx_user_kernel_compile": [C.c_char_p, C.c_char_p, C.POINTER(vp)],
        "mcx_rccl_unique_id": [vp],
        "mcx_exchange_rccl_init": [vp, vp],
        "mcx_exchange_rccl_adopt": [vp, vp],
        "mcx_exchange_rccl_destroy": [vp],
        "mcx_exchange_rccl_info": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "mcx_debug_exchange": [vp],
        "mcx_debug_fill_slot": [vp, C.c_float],
        "mcx_debug_persist_deal": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), u32p, C.c_int],
        "mcx_debug_copy_bandwidth": [C.c_size_t, C.c_int, C.POINTER(C.c_double)],
        "mcx_debug_numerics": [C.c_int, C.c_int, u32p, u32p],
        "mcx_debug_murray_screen": [C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.POINTER(C.c_uint64)],
        "mcx_debug_normals": [C.c_uint32] * 6 + [C.c_int, fp],
        "mcx_debug_sqrt_sweep": [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), u32p],
    }
    for name, args in sig.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            if os.environ.get("MCX_LIBMCX"):  # an older build of the library under tools/persist_ab.py: what it lacks is not called
                continue
            raise
        f.argtypes = args
        f.restype = C.c_int
    L.mcx_last_error.restype = C.c_char_p
    L.mcx_last_error.argtypes = []
    if os.environ.get("MCX_LIBMCX"):
        # a substituted build (tools/persist_ab.py) must at least speak this binding's ABI: the structs above are laid out for it
        import sys
        print("mcpar_amd: using MCX_LIBMCX=%s instead of the package's libmcx.so" % p, file=sys.stderr)
        got = L.mcx_abi_version()
        if got != ABI_VERSION:
            raise McxError(-1, "MCX_LIBMCX=%s is ABI %d, this binding is for ABI %d" % (p, got, ABI_VERSION))
    _lib = L
    return L


def check(code):
    if code != 0:
        raise McxError(code, load().mcx_last_error().decode("utf-8", "replace"))
