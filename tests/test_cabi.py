"""The C-ABI library loads without a GPU and exports every symbol include/mcx.h declares; without a
device every compute entry point fails loudly (no CPU fallback).  No compute is attempted here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(mcx_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_header_symbols_are_exported():
    import mcpar_amd
    lib = mcpar_amd.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.mcx_abi_version() == 4


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under mcpar_amd/ or include/ may reference it"""
    bad = []
    for base in ("mcpar_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            for f in fn:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".hh", ".cc", ".cpp")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if not f.endswith(".py"):  # prose in comments may cite the oracle; code may not
                        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
                        txt = re.sub(r"//[^\n]*", "", txt)
                    if re.search(r"mcxo_|oracle_lib|libmcx_oracle|oracle/mcx_oracle|_ref/", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def _has_gpu():
    import ctypes as C
    import mcpar_amd
    n = C.c_int(0)
    try:
        return mcpar_amd.load().mcx_device_info(None, 0, C.byref(n), None) == 0
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device failure mode")
def test_fails_loudly_without_a_device():
    import mcpar_amd as M
    with pytest.raises(M.McxError) as ei:
        M.Engine(2, 4)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)
    with pytest.raises(M.McxError):
        M.vlfunc_eval(M.VL_ROSENBROCK1, 2, [[1.0, 1.0]])
    with pytest.raises(M.McxError):
        M.debug_numerics(0, [1, 2, 3])


def test_header_is_plain_c99(tmp_path):
    """the boundary is a C ABI: include/mcx.h must compile as C99 and link against libmcx.so"""
    import subprocess
    src = tmp_path / "cabi.c"
    src.write_text('#include "mcx.h"\nint main(void){ mcx_vlfunc f = {MCX_VL_ROSENBROCK1, 2, 0, 0, 0, 0}; (void)f;\n'
                   ' mcx_plan_item it; int n = 0; (void)it;\n'
                   ' if (mcx_plan(10, 60, 10, 0.9f, 1u, 0u, 1, 0, 1, 256, 0, 0, 0, 0, &n) != MCX_OK || n < 3) return 2;\n'
                   ' return mcx_abi_version() == MCX_ABI_VERSION ? 0 : 1; }\n')
    exe = tmp_path / "cabi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcx",
                           "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd")])
    assert subprocess.call([str(exe)]) == 0
