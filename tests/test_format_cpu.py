"""The text of a sample row: `mcpar_amd/csrc/fmt_g6.hpp` (used by the facade's MCout::output on the host and by the engine's
text kernels on the GPU) must give, for every float, the characters `std::ostream << float` gives with the stream defaults --
what src/mcout.cc:41-45 prints -- i.e. glibc's printf("%g").  Sampled here; `fmt_check all` (every bit pattern, a few
minutes) was run when the file was written: 4 294 967 296 values, 0 differences."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_formatter_equals_the_c_library(tmp_path):
    exe = str(tmp_path / "fmt_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "fmt_check.cc")])
    r = subprocess.run([exe, "text"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout
    a, b = r.stdout.splitlines()[:2]
    assert a == b and a.startswith("1e-05  123456  1e+10  -0  0.1  1.23457e+06  3.14159  -1.5e-07  inf  ")
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith(", 0 differences")
