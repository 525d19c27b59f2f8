"""MCout of the C++ facade on the CPU (no GPU, no MPI): the reference's storage semantics and text
format (src/mcout.cc:30-48 two spaces after every field, ostream default precision; :52-94 collect;
:129-145 add / running maximum)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mcout_semantics_and_format(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "mcpar_amd", "drivers"), "../libmcpar.so"],
                          stdout=subprocess.DEVNULL)
    exe = str(tmp_path / "mcout_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "mcout_check.cc"), "-o", exe,
                           "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcpar", "-lmcx",
                           "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout == ("size 2 maxsize 3 ncol 3 vsize 9\n"
                          "1.5  -2  -3.25  \n"
                          "0.1  1e-05  0.5  \n"
                          "123456  1e+10  0.25  \n"
                          "--\n"
                          "collect 9 -3.25 0.5 0.25\n"
                          "collect again 0 1\n"
                          "maxlike 0.5 0.1 1e-05\n"
                          "getpset 0.1 0.5 0.25\n")
