"""The reference-shaped C++ surface (include/mcpar/*.hh, libmcpar.so) and the demo drivers:
same command lines, same stdout/file formats as the reference drivers, values equal
to the oracle's."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "mcpar_amd", "drivers")


def fmt_rows(rows):
    """MCout::output: every field followed by two spaces, ostream default (%g, 6 digits)"""
    return "".join("".join("%g  " % v for v in r) + "\n" for r in rows)


def build_drivers():
    subprocess.check_call(["make", "-C", DRV], stdout=subprocess.DEVNULL)


def test_mcpar_dgauss_driver_output(tmp_path):
    build_drivers()
    r = subprocess.run([os.path.join(DRV, "mcpar-dgauss")], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vl, keep = O.make_vlfunc(O.VL_DUALGAUSS, 2, [5.0])
    e = O.Engine(2, 4)  # MCPar(nparam, 4, size, rank) defaults, src/mcpar-dgauss.cc:31
    e.run(8, 500, np.array([0, 0, 2, 2, 0, 1.5, 0, -2], np.float32), vl)
    s = e.samples
    best = int(np.argmax(s[:, 2]))
    assert np.all(s[best, 2] >= s[:, 2])
    expect = fmt_rows(s) + "max likelihood value: %g\n" % s[best, 2] + "%g  %g  \n" % (s[best, 0], s[best, 1])
    assert r.stdout == expect
    assert len(r.stdout.splitlines()) == 34  # 32 sample rows + 2
    # per-rank parameter file: tab separated, trailing tab (src/mcpar-dgauss.cc:38-47)
    txt = (tmp_path / "mcpar-dgauss.000.txt").read_text()
    assert txt == "".join("%g\t%g\t\n" % (a, b) for a, b in s[:, :2])
    log = (tmp_path / "mcpar-log.000.txt").read_text()
    assert log.startswith("Starting burn-in.  Samples = 500\nStarting main sample loop:  nsamp = 8\n"
                          "Output after each 5 steps.\nBeginning output at step 5\nOutput finished\n")


def test_mcpar_rosen1_driver_output(tmp_path):
    build_drivers()
    r = subprocess.run([os.path.join(DRV, "mcpar-rosen1"), "120"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, 2)
    e = O.Engine(2, 4)
    e.run(120, 500, np.array([0, 0, 2, 2, 0, 1.5, 0, -2], np.float32), vl)
    assert r.stdout == "nsamp = 120\n" + fmt_rows(e.samples)
    log = (tmp_path / "mcpar-log.000.txt").read_text()
    assert log.count("Beginning output at step") == 9 and "Output after each 12 steps." in log


def test_cpp_api_builtin_and_user_vlfunc(tmp_path):
    build_drivers()
    exe = str(tmp_path / "facade_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_check.cc"), "-o", exe,
                           "-L", os.path.join(ROOT, "mcpar_amd"), "-lmcpar", "-lmcx",
                           "-Wl,-rpath," + os.path.join(ROOT, "mcpar_amd")])
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a, b, tail = r.stdout.split("=====\n")
    assert a == b  # a user VLFunc subclass (host callback) reproduces the fused device run bit for bit
    np_, nc, nsamp, nburn = 8, 48, 60, 120
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, np_)
    e = O.Engine(np_, nc, pl=0.8)
    e.run(nsamp, nburn, O.default_pinit(np_, nc), vl)
    s = e.samples
    best = int(np.argmax(s[:, np_]))
    expect = fmt_rows(s) + "size %d maxsize %d ncol %d maxlike %g p0 %g accepts %d\n" % (
        nc * nsamp, nc * nsamp, np_ + 1, s[best, np_], s[best, 0], e.naccept_main)
    assert a == expect
    # VLFunc is called once before burn-in and once per step (src/mcpar.cc:53,60,160)
    assert "user calls %d" % (1 + nburn + nsamp) in tail
    # ... and, on request, once more per chain after every main-loop step, result discarded, like the reference
    # (src/mcpar.cc:177-182: MCX_OPT_REFERENCE_CALLS / MCPAR_REFERENCE_CALLS=1) -- same rows, more calls
    r2 = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=dict(os.environ, MCPAR_REFERENCE_CALLS="1"))
    assert r2.returncode == 0, r2.stderr
    a2, b2, tail2 = r2.stdout.split("=====\n")
    assert a2 == a and b2 == b
    assert "user calls %d" % (1 + nburn + nsamp + nsamp * nc) in tail2
    assert "guard: N for Rosenbrock1 must be even and >= 2" in tail
    # mcpar.logging = true, logstep = 7, nsamp = 20 (outstep 5), 4 chains: the reference's diagnostics, in
    # its order (src/mcpar.cc:115-126): dump message of iteration i, then the diagnostic of iteration i
    log = (tmp_path / "mcpar-log.000.txt").read_text()
    want = ("Starting burn-in.  Samples = 10\nStarting main sample loop:  nsamp = 20\nOutput after each 5 steps.\n"
            "sample step 0:\toutsamples size= 0  maxsize = 80  ncol= 9\n\tvsize = 720  offset = 0\n"
            "Beginning output at step 5\nOutput finished\n\n"
            "sample step 7:\toutsamples size= 28  maxsize = 80  ncol= 9\n\tvsize = 720  offset = 63\n"
            "Beginning output at step 10\nOutput finished\n\n"
            "sample step 14:\toutsamples size= 56  maxsize = 80  ncol= 9\n\tvsize = 720  offset = 126\n"
            "Beginning output at step 15\nOutput finished\n\n")
    assert want in log, log
    assert "r2 0 -1" in tail


def test_mcpar_run_driver_flags(tmp_path):
    build_drivers()
    r = subprocess.run([os.path.join(DRV, "mcpar-run"), "--func", "mix", "--np", "32", "--nc", "256", "--nsamp", "40",
                        "--nburn", "60", "--pl", "0.85", "--quiet"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "" and "accept rate (main)" in r.stderr and "max likelihood value" in r.stderr
    r = subprocess.run([os.path.join(DRV, "mcpar-run"), "--func", "rosen1", "--np", "3"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "must be even" in r.stderr


def test_mcpar_run_binary_rows_equal_text_rows(tmp_path):
    """--binary: the same rows in the same order as the text dump, as raw float32 (text formatting is what the
    reference spends 65-87 % of its wall time on); the samples stream through the engine's sink, nsamp/10 steps
    per block, the last block ragged"""
    build_drivers()
    args = ["--func", "rosen1", "--np", "8", "--nc", "96", "--nsamp", "73", "--nburn", "60", "--pl", "0.9"]
    t = subprocess.run([os.path.join(DRV, "mcpar-run")] + args, cwd=tmp_path, capture_output=True, timeout=300)
    b = subprocess.run([os.path.join(DRV, "mcpar-run")] + args + ["--binary"], cwd=tmp_path, capture_output=True, timeout=300)
    assert t.returncode == 0 and b.returncode == 0, (t.stderr, b.stderr)
    rows = np.frombuffer(b.stdout, dtype="<f4").reshape(-1, 9)
    assert rows.shape[0] == 73 * 96
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, 8)
    eo = O.Engine(8, 96, pl=0.9)
    eo.run(73, 60, O.default_pinit(8, 96), vl)
    assert np.array_equal(rows.view(np.uint32), eo.samples.view(np.uint32))
    assert t.stdout.decode() == fmt_rows(eo.samples)
    # --stream-text (MCout::text_only): the same bytes, formatted on the GPU and never stored on the host; same maximum
    st = subprocess.run([os.path.join(DRV, "mcpar-run")] + args + ["--stream-text"], cwd=tmp_path, capture_output=True, timeout=300)
    assert st.returncode == 0, st.stderr
    assert st.stdout == t.stdout
    assert st.stderr.decode().split("max likelihood value:")[1] == t.stderr.decode().split("max likelihood value:")[1]
    # --iter: the iteration index the R analysis script reconstructs (src/anly/mcpar-analysis.R:80-120) in front of every row
    it = subprocess.run([os.path.join(DRV, "mcpar-run")] + args + ["--iter"], cwd=tmp_path, capture_output=True, timeout=300)
    assert it.returncode == 0, it.stderr
    want = fmt_rows(eo.samples).splitlines()
    assert it.stdout.decode().splitlines() == ["%d  %s" % (r // 96, line) for r, line in enumerate(want)]


def test_mcpar_run_with_the_users_own_likelihood_as_source(tmp_path):
    """mcpar-run --func-source FILE: SourceVLFunc (include/mcpar/vlfunc.hh) -> MCX_VL_SOURCE, the user's device functions
    compiled into the fused step kernels.  The example restates Rosenbrock1: the driver's text must be the built-in's, byte
    for byte; a text that does not compile ends the run with the compiler's message (the reference abort()s on library
    failures: src/mcpar.hh:93)"""
    build_drivers()
    args = ["--np", "16", "--nc", "200", "--nsamp", "40", "--nburn", "120", "--pl", "0.9"]
    ex = os.path.join(ROOT, "mcpar_amd", "examples")
    ref = subprocess.run([os.path.join(DRV, "mcpar-run"), "--func", "rosen1"] + args, cwd=tmp_path, capture_output=True, timeout=300)
    assert ref.returncode == 0, ref.stderr
    for name, par in (("user_rosenbrock1_blocks.hip", []), ("user_rosenbrock1_whole.hip", ["--par", "1.0"])):
        r = subprocess.run([os.path.join(DRV, "mcpar-run"), "--func-source", os.path.join(ex, name)] + par + args, cwd=tmp_path,
                           capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stdout == ref.stdout and len(r.stdout) > 100000, name
    bad = tmp_path / "bad.hip"
    bad.write_text("__device__ float mcx_user_loglike(const float *x, int d, const float *par) { return x[0] +; }\n")
    r = subprocess.run([os.path.join(DRV, "mcpar-run"), "--func-source", str(bad)] + args, cwd=tmp_path, capture_output=True, timeout=300)
    assert r.returncode != 0 and b"does not compile" in r.stderr and b"mcx_user_likelihood:1" in r.stderr


def test_mcpar_run_out_file_takes_every_kind_of_dump(tmp_path):
    """--out FILE (MCout::text_file): the file gets what stdout would have got -- also when the rows take MCout::output()'s
    own path (--iter keeps them for the driver, so run() dumps through output()) and under --binary (the file
    used to stay empty while the rows went to stdout)"""
    build_drivers()
    args = ["--func", "rosen1", "--np", "8", "--nc", "64", "--nsamp", "30", "--nburn", "60", "--pl", "0.9"]
    t = subprocess.run([os.path.join(DRV, "mcpar-run")] + args, cwd=tmp_path, capture_output=True, timeout=300)
    b = subprocess.run([os.path.join(DRV, "mcpar-run")] + args + ["--binary"], cwd=tmp_path, capture_output=True, timeout=300)
    assert t.returncode == 0 and b.returncode == 0
    for extra, want in (([], t.stdout), (["--binary"], b.stdout)):
        f = tmp_path / ("out" + "".join(extra))
        r = subprocess.run([os.path.join(DRV, "mcpar-run")] + args + extra + ["--out", str(f)], cwd=tmp_path, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stdout == b"" and f.read_bytes() == want, extra


MPIEXEC = "/opt/conda/bin/mpiexec"


@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI launcher in this image")
def test_two_mpi_ranks_through_the_facade(tmp_path):
    """the reference's process model: mpiexec -n 2, one rank per shard (both on this one GPU), the
    MPI_Allgather of src/mcpar.cc:127-140 staged through the exchange hook; stdout = MCout::output's
    rank-major dumps (src/mcout.cc:62-69) and must equal the oracle's two-shard run"""
    r = subprocess.run(["make", "-C", DRV, "mpi"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("MPI build not available: " + r.stderr[-300:])
    np_, nc, nsamp, nburn, pl = 8, 32, 40, 60, 0.8
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([MPIEXEC, "-n", "2", os.path.join(DRV, "mcpar-run-mpi"), "--func", "rosen1", "--np", str(np_),
                        "--nc", str(nc), "--nsamp", str(nsamp), "--nburn", str(nburn), "--pl", str(pl)],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    # one GPU: the two ranks share it, RCCL refuses such a communicator and the facade falls back (collectively)
    # to the host-staged MPI_Allgather; with two GPUs the ranks bind to one each and the exchange is RCCL
    import mcpar_amd as M
    assert ("exchange: rccl" if M.engine.device_count() >= 2 else "exchange: mpi-staged") in r.stderr, r.stderr[-500:]
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, np_)
    engs = [O.Engine(np_, nc, nshards=2, shard=s, pl=pl) for s in range(2)]
    O.run_all(engs, nsamp, nburn, [O.default_pinit(np_, nc, g0=s * nc) for s in range(2)], vl)
    assert engs[0].remote_steps > 0
    outstep = nsamp // 10 if nsamp > 50 else 5
    bounds = [s for s in range(outstep, nsamp, outstep)] + [nsamp]
    expect, lo = "", 0
    for hi in bounds:  # each dump: rank 0's new rows, then rank 1's
        for e in engs:
            expect += fmt_rows(e.samples[lo * nc:hi * nc])
        lo = hi
    assert r.stdout == expect
    # --stream-text on two ranks: every rank's blocks as text from its GPU, written by rank 0 in rank order: the same bytes
    st = subprocess.run([MPIEXEC, "-n", "2", os.path.join(DRV, "mcpar-run-mpi"), "--func", "rosen1", "--np", str(np_),
                         "--nc", str(nc), "--nsamp", str(nsamp), "--nburn", str(nburn), "--pl", str(pl), "--stream-text"],
                        cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert st.returncode == 0, st.stderr[-2000:]
    assert st.stdout == expect
    assert st.stderr.split("max likelihood value:")[1] == r.stderr.split("max likelihood value:")[1]


@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI launcher in this image")
@pytest.mark.parametrize("stream", [True, False], ids=["text-only", "rows-and-text"])
def test_two_mpi_ranks_write_their_text_side_by_side(tmp_path, stream):
    """MCout::text_file / mcpar-run --out: every rank writes its share of a dump itself, at the byte offset an MPI_Exscan
    of the shares' sizes gives it; the file must be byte for byte what the funnel through rank 0 prints"""
    r = subprocess.run(["make", "-C", DRV, "mpi"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("MPI build not available: " + r.stderr[-300:])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [MPIEXEC, "-n", "2", os.path.join(DRV, "mcpar-run-mpi"), "--func", "rosen1", "--np", "8", "--nc", "640",
            "--nsamp", "60", "--nburn", "60", "--pl", "0.9"] + (["--stream-text"] if stream else [])
    funnel = subprocess.run(base, cwd=tmp_path, capture_output=True, timeout=300, env=env)
    assert funnel.returncode == 0, funnel.stderr[-2000:]
    out = tmp_path / "samples.txt"
    side = subprocess.run(base + ["--out", str(out)], cwd=tmp_path, capture_output=True, timeout=300, env=env)
    assert side.returncode == 0, side.stderr[-2000:]
    assert side.stdout == b""  # everything went to the file
    got = out.read_bytes()
    assert len(got) > 100000 and got == funnel.stdout
    # and it is the oracle's two-shard run in dump order (each dump: rank 0's new rows, then rank 1's)
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, 8)
    engs = [O.Engine(8, 640, nshards=2, shard=s, pl=0.9) for s in range(2)]
    O.run_all(engs, 60, 60, [O.default_pinit(8, 640, g0=s * 640) for s in range(2)], vl)
    expect, lo = "", 0
    for hi in list(range(6, 60, 6)) + [60]:
        for e in engs:
            expect += fmt_rows(e.samples[lo * 640:hi * 640])
        lo = hi
    assert got.decode() == expect
