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


# ---------------------------------------------------------------------------------------------
# Pinned by the reference itself: tests/golden/mcout_reference.json holds scripts and the transcripts the
# REFERENCE's MCout (src/mcout.cc compiled in place with the real MPI headers, oracle/Makefile) produced for
# them on 1 rank and under `mpiexec -n 2` (oracle/gen_golden.py).  The same driver source, built against
# the facade's header, must print the same transcripts byte for byte: text format of output()
# (src/mcout.cc:30-48), incremental dumps and rewind, collect()'s rank-major order (:52-94), maxlike with
# its first-maximum / lowest-rank tie rules (:96-127, :140-144), NaN and -inf rows.
# ---------------------------------------------------------------------------------------------
import json  # noqa: E402

import pytest  # noqa: E402

MPI = "/opt/conda"
MPIEXEC = os.path.join(MPI, "bin", "mpiexec")
DRV = os.path.join(ROOT, "mcpar_amd", "drivers")


def _scenarios():
    with open(os.path.join(ROOT, "tests", "golden", "mcout_reference.json")) as f:
        return json.load(f)["scenarios"]


def _write_scripts(tmp_path, sc):
    files = []
    for r, lines in enumerate(sc["scripts"]):
        fn = tmp_path / ("%s.%d" % (sc["name"], r))
        fn.write_text("\n".join(lines) + "\n")
        files.append(str(fn))
    return files


def test_facade_mcout_equals_reference_mcout_on_one_rank(tmp_path):
    subprocess.check_call(["make", "-C", DRV, "../libmcpar.so"], stdout=subprocess.DEVNULL)
    exe = str(tmp_path / "mcout_script")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include", "mcpar"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mcout_script.cc"),
                           "-o", exe, "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcpar", "-lmcx",
                           "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd")])
    seen = 0
    for sc in _scenarios():
        if sc["nranks"] != 1:
            continue
        out = subprocess.run([exe, str(sc["np"])] + _write_scripts(tmp_path, sc), capture_output=True, text=True,
                             timeout=60)
        assert out.returncode == 0, out.stderr
        assert out.stdout == sc["transcript"], sc["name"]
        seen += 1
    assert seen >= 3


def test_bulk_output_prints_what_the_stream_prints(tmp_path):
    """the threaded fmtg6 path of MCout::output (1.5 M numbers of every kind) against `ostream << float`, and a stream with
    its own precision, which the facade must honour"""
    subprocess.check_call(["make", "-C", DRV, "../libmcpar.so"], stdout=subprocess.DEVNULL)
    exe = str(tmp_path / "mcout_bulk")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include", "mcpar"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mcout_bulk.cc"),
                           "-o", exe, "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcpar", "-lmcx",
                           "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI launcher in this image")
def test_facade_mcout_equals_reference_mcout_on_two_ranks(tmp_path):
    r = subprocess.run(["make", "-C", DRV, "../libmcpar_mpi.so", "MPI=" + MPI], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("MPI build of the facade not available: " + r.stderr[-300:])
    exe = str(tmp_path / "mcout_script_mpi")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DMCX_WITH_MPI", "-DMPICH_SKIP_MPICXX", "-DOMPI_SKIP_MPICXX",
                           "-I", os.path.join(ROOT, "include", "mcpar"), "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(MPI, "include"), os.path.join(ROOT, "tests", "cpp", "mcout_script.cc"),
                           "-o", exe, "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcpar_mpi", "-lmcx",
                           os.path.join(MPI, "lib", "libmpi.so"), "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd"),
                           "-Wl,-rpath,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath," + os.path.join(MPI, "lib")])
    seen = 0
    for sc in _scenarios():
        out = subprocess.run([MPIEXEC, "-n", str(sc["nranks"]), exe, str(sc["np"])] + _write_scripts(tmp_path, sc),
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout == sc["transcript"], sc["name"]
        seen += sc["nranks"] == 2
    assert seen >= 2


@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI launcher in this image")
@pytest.mark.parametrize("nranks", [1, 3])
def test_text_written_side_by_side_equals_the_funnel(tmp_path, nranks):
    """MCout::write_text both ways (tests/cpp/textfile_check.cc): through rank 0's stream, and every rank writing its own
    share of every block into one file at an MPI_Exscan'd offset (MCout::text_file) -- the same bytes, in dump order"""
    r = subprocess.run(["make", "-C", DRV, "../libmcpar_mpi.so", "MPI=" + MPI], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("MPI build of the facade not available: " + r.stderr[-300:])
    exe = str(tmp_path / "textfile_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DMCX_WITH_MPI", "-DMPICH_SKIP_MPICXX", "-DOMPI_SKIP_MPICXX",
                           "-I", os.path.join(ROOT, "include", "mcpar"), "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(MPI, "include"), os.path.join(ROOT, "tests", "cpp", "textfile_check.cc"),
                           "-o", exe, "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcpar_mpi", "-lmcx",
                           os.path.join(MPI, "lib", "libmpi.so"), "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd"),
                           "-Wl,-rpath,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath," + os.path.join(MPI, "lib")])
    out = subprocess.run([MPIEXEC, "-n", str(nranks), exe, str(tmp_path / "side_by_side.txt")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    # output() under text_file() -- the row path, as text between write_text blocks and as raw rows under binary() -- lands in
    # the file, in dump order, and nowhere else (it used to go to the stream and leave the file short)
    assert lines[0].startswith("rows same ") and lines[1].startswith("binary same %d bytes" % (nranks * (7 + 8 + 9) * 12)), out.stdout
    assert lines[2] == "funnel in dump order" and lines[3].startswith("same ") and lines[3].endswith("%d ranks" % nranks), out.stdout
    assert len(lines) == 4, out.stdout


def test_percent_g_is_the_reference_row_format():
    """tests/test_gpu_facade.py formats oracle rows with '%g  ' to build its expectations; the reference's own
    text for the golden rows says that this is the format of src/mcout.cc:41-45 (nan, -inf, -0, 1e-05, 1e+10, ...)."""
    import struct
    sc = [s for s in _scenarios() if s["name"] == "one_rank_formats"][0]
    rows = [[struct.unpack("<f", struct.pack("<I", int(w, 16)))[0] for w in ln.split()[1:]]
            for ln in sc["scripts"][0] if ln.startswith("add")]
    text = [ln[4:-1] for ln in sc["transcript"].splitlines() if ln.startswith("r0 |")][-len(rows):]
    assert len(rows) == 10 and text == ["".join("%g  " % v for v in r) for r in rows]
