#!/usr/bin/env python3
"""bench.py -- chain-steps/sec of the MI355X-native Metropolis-Hastings engine (libmcx, C ABI) on the
configurations of BASELINE.json / SURVEY §8d.

Default workload (`--config c3`, the one BASELINE.json's metric is quoted on): Rosenbrock1(16) x 65 536
chains per GPU, R-local job (pl = 1.0, nburn = 500, nsamp = 1000, default tuner constants, seed 8675309,
pinit[g][i] = 0.5 sin(0.37 (g d + i))), every (chain, step) sample row retained in HBM like the reference's
MCout.  Other configurations: c2 (8-D x 4096, R-local), c5 (32-D 8-component mixture x 32 768 per GPU,
R-murray: pl = 0.9, nburn = 500, nsamp = 100), c3-murray (C3's shape, R-murray).

A bench "step" is ONE complete MCPar::run-shaped job (burn-in with tuner + main loop with sample emission
+ the inter-shard exchange when N > 1) = n (nburn + nsamp) chain-steps per GPU.  Chains shard across GPUs
(weak scaling: the configuration's chain count per GPU), the (mu, sig^2) slots are exchanged with the
library's own in-place ncclAllGather (RCCL over xGMI, mcx_exchange_rccl_*).

  python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c5|c3-murray]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  At N = 1 the line also carries: `end_to_end` (the same job with every sample
row copied out to host memory), `roofline` with hardware counters collected live by child `rocprofv3 --pmc`
passes of this same script -- of the kernel that actually ran: its name is taken from the child's own kernel
trace, and without live counters `frac` is null --, the other configurations as `config.other_configs`,
`host_callback` (the same shape with a user likelihood on the host), `cpu_baseline` (the CPU oracle on the
host cores) and, as the LAST key, `summary`: every headline number again in a few hundred bytes, because
whoever keeps only the tail of a long line should still see them.  What each field means and how it is
computed is written down once, in DESIGN.md section 6 ("the bench line"), not in the line.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver on this pool only supports dmabuf IPC (needed by RCCL across processes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PORT_OVER_REFERENCE = 2.2  # the oracle port's speed over the compiled reference per set of cores
HBM_PEAK = 8.0e12       # B/s, MI355X_MICROARCH.md chip table
FP32_VALU_PEAK = 157.3e12  # flop/s, MI355X_MICROARCH.md "Peak FP32 (vector)"
BF16_MFMA_PEAK = 2.5e15    # flop/s dense, MI355X_MICROARCH.md (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD)
N_SIMD = 1024           # 256 CUs x 4 SIMDs

# SURVEY §8d.  lik: 1 = Rosenbrock1, 5 = K-component unit-variance Gaussian mixture
CONFIGS = {
    "c2": dict(name="mcpar-rosen1 (C2)", lik=1, d=8, n=4096, pl=1.0, nburn=500, nsamp=1000),
    "c3": dict(name="mcpar-rosen2 (C3)", lik=1, d=16, n=65536, pl=1.0, nburn=500, nsamp=1000),
    "c5": dict(name="mcpar-dgauss mixture (C5 per-GPU shape)", lik=5, d=32, K=8, n=32768, pl=0.9, nburn=500, nsamp=100),
    "c3-murray": dict(name="mcpar-rosen2 (C3), R-murray", lik=1, d=16, n=65536, pl=0.9, nburn=500, nsamp=100),
    # flagged variant, not reference behaviour: the overlapping Rosenbrock with the sign of
    # src/rosenbrock.cc:38 corrected and the loop kept inside each set (MCX_VL_ROSENBROCK2_FIXED)
    "c3-rosen2fixed": dict(name="mcpar-rosen2 (C3) with the well-posed overlapping Rosenbrock2 (flagged deviation)", lik=6, d=16,
                           n=65536, pl=1.0, nburn=500, nsamp=1000),
}


def alg_bytes_per_chain_step(d, main, emit):
    """state round trip 2*4*(3d+1) main / 2*4*(d+1) burn-in, + 4(d+1) per kept sample"""
    b = 8 * (3 * d + 1) if main else 8 * (d + 1)
    return b + (4 * (d + 1) if (main and emit) else 0)


def pinit_for(d, n, g0):
    import numpy as np
    g = np.arange(g0, g0 + n, dtype=np.float64)[:, None]
    i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)


def mix_params(d, K):
    """K unit-variance components, means 5k/(K-1) * 1, weights (5, 1, ..., 1)"""
    import numpy as np
    means = np.stack([np.full(d, 5.0 * k / (K - 1)) for k in range(K)]).astype(np.float32)
    return np.concatenate([means.ravel(), [5] + [1] * (K - 1)]).astype(np.float32)


def lpc_for(d):
    """lanes per chain: next power of two of ceil(d / 4) (mcx_engine.hip:lpc_for)"""
    nb, l = (d + 3) // 4, 1
    while l < nb:
        l <<= 1
    return l


def make_lik(mod, cfg):
    if cfg["lik"] == 5:
        return mod.make_vlfunc(mod.VL_GAUSSMIX, cfg["d"], mix_params(cfg["d"], cfg["K"]), cfg["K"])
    if cfg["lik"] == 6:
        return mod.make_vlfunc(6, cfg["d"])  # MCX_VL_ROSENBROCK2_FIXED (same id in the oracle)
    return mod.make_vlfunc(mod.VL_ROSENBROCK1, cfg["d"])


def workload_text(cfg, n):
    lik = ("Rosenbrock1(%d)" % cfg["d"] if cfg["lik"] == 1 else "Rosenbrock2Fixed(%d)" % cfg["d"] if cfg["lik"] == 6 else
           "%d-D %d-component Gaussian mixture" % (cfg["d"], cfg["K"]))
    return ("%s: %s x %d chains/GPU, %s job (pl=%.2f, nburn=%d, nsamp=%d, sync=10), one bench step = one full run()"
            % (cfg["name"], lik, n, "R-local" if cfg["pl"] >= 1.0 else "R-murray", cfg["pl"], cfg["nburn"], cfg["nsamp"]))


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a port of the reference algorithm) on this box's host cores
# ---------------------------------------------------------------------------------------------
def cpu_baseline(cfg):
    """Reported beside the GPU number; never part of the product path.  R-local configurations run the
    whole job; R-murray ones a bounded sample (the O(n N d) sweeps cost the CPU minutes at full size)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the GPU box's CPU share for one GPU is 16 cores
    d, n, nburn, nsamp, pl = cfg["d"], cfg["n"], cfg["nburn"], cfg["nsamp"], cfg["pl"]
    sample = "the whole job"
    if pl < 1.0:
        n = min(n, 8192)
        sample = "same job on %d chains (the Murray sweep is O(n N d) per pass)" % n
    vl, _keep = make_lik(O, cfg)
    p = pinit_for(d, n, 0)
    e = O.Engine(d, n, pl=pl, threads=cores)
    e.set_record(samples=True, mask=False)
    t0 = time.perf_counter()
    e.run(nsamp, nburn, p, vl)
    dt = time.perf_counter() - t0
    passes = e.remote_passes
    e.close()
    v = n * (nburn + nsamp) / dt
    return dict(value=v, unit="chain-steps/s", cores=cores, kind="port",
                sample="%s: %d chains x %d-D, %d burn-in + %d main steps, pl=%.2f (%d Murray passes), samples kept in "
                       "host memory, OpenMP over chains (%.1f s of %d cores)" % (sample, n, d, nburn, nsamp, pl, passes, dt, cores),
                reference_equivalent=dict(value=v / PORT_OVER_REFERENCE, unit="chain-steps/s", port_over_reference=PORT_OVER_REFERENCE))


# ---------------------------------------------------------------------------------------------
# hardware counters: child rocprofv3 --pmc passes of this script (one counter group per pass)
# ---------------------------------------------------------------------------------------------
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"), ("GRBM_GUI_ACTIVE",),
              ("SQ_INSTS_VALU_FLOPS_FP32", "SQ_INSTS_VALU_IOPS", "SQ_INSTS_VALU_TRANS_F32"),
              ("SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32"))


def pmc_child(cfg, n):
    """what runs under rocprofv3: one warm job and one measured job of the configuration, nothing else"""
    import mcpar_amd as M
    M.load().mcx_set_device(0)
    eng = M.Engine(cfg["d"], n, pl=cfg["pl"])
    vl, _keep = make_lik(M, cfg)
    eng.stage_pinit(pinit_for(cfg["d"], n, 0))
    for _ in range(2):
        eng.run(cfg["nsamp"], cfg["nburn"], None, vl)
    eng.close()


PMC_PASS_MFMA = ("SQ_VALU_MFMA_BUSY_CYCLES",)  # Murray configurations: the screen's matrix-core kernel (cycles, not quad-cycles)


def collect_pmc(config_name, n, keep_dir=None, budget_s=150.0, extra_passes=()):
    """-> ({kernel: {counter: mean per dispatch}}, {kernel: mean duration ns under the profiler}, note).
    Each counter group is its own `rocprofv3 --pmc ... --kernel-trace` run of `bench.py --pmc-child`."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, None, "rocprofv3 not on PATH"
    out, dur, t_start = {}, {}, time.time()
    if keep_dir:
        tmp = os.path.abspath(keep_dir)
        os.makedirs(tmp, exist_ok=True)
    else:
        tmp = tempfile.mkdtemp(prefix="mcx_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    for group in tuple(PMC_PASSES) + tuple(extra_passes):
        if time.time() - t_start > budget_s:
            return out or None, dur, "time budget exhausted before " + ",".join(group)
        ddir = os.path.join(tmp, "_".join(group))
        cmd = [exe, "--pmc", *group, "--kernel-trace", "-f", "csv", "-d", ddir, "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", config_name, "--chains", str(n)]
        try:
            r = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=90,
                               env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")))
        except Exception as ex:  # noqa: BLE001
            return out or None, dur, "rocprofv3 %s: %r" % (",".join(group), ex)
        files = glob.glob(os.path.join(ddir, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return out or None, dur, "rocprofv3 %s failed (rc %d): %s" % (",".join(group), r.returncode, r.stdout[-300:].decode("utf-8", "replace"))
        acc, kd = {}, {}
        for row in csv.DictReader(open(files[0])):
            acc.setdefault((row["Kernel_Name"], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            if row["Counter_Name"] == group[0] and row.get("End_Timestamp") and row.get("Start_Timestamp"):
                kd.setdefault(row["Kernel_Name"], []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        for (k, c), v in acc.items():
            # the second job's dispatches only (the first job warms allocations and code objects)
            v = v[len(v) // 2:]
            out.setdefault(k, {})[c] = sum(v) / len(v)
        for k, v in kd.items():  # dispatch durations under the profiler, of this pass: (mean ns, dispatches of the measured job)
            v = v[len(v) // 2:]
            dur.setdefault(k, {})["_".join(group)] = (sum(v) / len(v), len(v))
    if not keep_dir:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, dur, None


def find_kernel(table, *needles):
    for k in table or {}:
        if all(s in k for s in needles):
            return k
    return None


def dominant_kernel(pmc, pdur, family):
    """the kernel of `family` (a name prefix: k_fused_fast covers k_fused_fastb, k_run_small its instantiations) that the
    measured child job spent the most time in, by the child's own kernel trace: (name as rocprofv3 reports it, launches of
    the measured job, mean duration ns)"""
    import re

    def main_loop(k):  # k_fused_fast<LPC, MAIN, ...> / k_fused_fastb<LPC2, BPL, MAIN, ...>: the main-loop instantiation
        m = re.search(r"k_fused_fastb?<([^>]*)>", k)
        if not m:
            return True
        a = [x.strip() for x in m.group(1).split(",")]
        return (a[2] if "k_fused_fastb<" in k else a[1]) == "true"
    best = None
    for want_main in (True, False):  # (the figures beside the name are the main-loop launches')
        for k, by_pass in (pdur or {}).items():
            if family not in k or not by_pass or (want_main and not main_loop(k)):
                continue
            mean_ns = sum(v[0] for v in by_pass.values()) / len(by_pass)
            count = max(v[1] for v in by_pass.values())
            if best is None or mean_ns * count > best[1] * best[2]:
                best = (k, mean_ns, count)
        if best:
            break
    return best


def job_stats(eng, runs=1):
    """what the engine says about its last run (mcx_counters): launches, exchanges and the time the step stream waited for
    them, tuner meetings of the one-launch small-n kernel that were abandoned (over the engine's life)"""
    c = eng.counters
    return dict(kernel_launches_per_run=c["kernel_launches"], exchanges_per_run=c["exchanges"],
                exchange_waits_per_run=c["exchange_waits"], exchange_wait_ms_per_run=c["exchange_wait_ns"] / 1e6,
                meet_timeouts=c["meet_timeouts_total"], small_n_launches_per_run=c["small_n_launches"],
                small_n_blocks_per_lane=c["small_n_blocks_per_lane"])


# ---------------------------------------------------------------------------------------------
# N = 1 extras that put DESIGN.md's remaining claims under the driver's clock
# ---------------------------------------------------------------------------------------------
def spd_covariance(d):
    """a fixed, well-conditioned full covariance: 0.01 (I + 0.5 a a^T / d) with a from a seeded generator"""
    import numpy as np
    a = np.random.default_rng(1234 + d).normal(size=(d, d))
    return (0.01 * (np.eye(d) + 0.5 * a @ a.T / d)).astype(np.float32)


def time_job(eng, vl, p, nsamp, nburn, incov=None, reps=5):
    """median wall time of `reps` runs after warm runs -- at least two (run() returns when the stream has drained; the first
    run of a kernel family also loads its code object) and 20 ms' worth: the first jobs after a pause of the GPU run at the
    clock it idled at (a 0.36 ms job measures 0.39 there)"""
    eng.stage_pinit(p)
    t_warm, k = time.perf_counter(), 0
    while k < 2 or (time.perf_counter() - t_warm < 0.02 and k < 200):
        eng.run(nsamp, nburn, None, vl, incov)
        k += 1
    ts = []
    for r in range(reps):
        t0 = time.perf_counter()
        eng.run(nsamp, nburn, None, vl, incov)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts)
    return ts[len(ts) // 2]


def time_jobs_in_turn(jobs, reps=7):
    """median wall times of several kinds of job run IN TURN (a, b, a, b, ...) after two warm rounds: what a ratio of two job
    times should be taken from -- the GPU's clock drifts with what it has just been doing (the same 0.4 ms job measured
    0.405 and 0.430 ms in two bench runs a minute apart), and jobs timed minutes apart carry that drift into their ratio"""
    ts = [[] for _ in jobs]
    for r in range(reps + 2):
        for i, job in enumerate(jobs):
            t0 = time.perf_counter()
            job()
            ts[i].append(time.perf_counter() - t0)
    return [sorted(t[2:])[len(t[2:]) // 2] for t in ts]


def claims_under_the_clock(M, E, headline_ms, headline_n, nburn, nsamp):
    """full-covariance cost against diagonal (16-D, 32-D), the strong-scaling proxy (8192 x 16-D), and the accept
    rate of the shape the survey measured on the compiled reference (100 + 20 steps)"""
    out = {}
    fc = {}
    for d in (16, 32):
        n = 65536
        vl, _k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
        p = pinit_for(d, n, 0)
        eng = M.Engine(d, n, pl=1.0)
        eng.set_option(E.OPT_SAMPLES, 0)  # summary only: the 32-D rows of 1000 steps would be 8.7 GB
        eng.stage_pinit(p)
        cov = spd_covariance(d)
        t_diag, t_full = time_jobs_in_turn([lambda: eng.run(nsamp, nburn, None, vl), lambda: eng.run(nsamp, nburn, None, vl, cov)])
        eng.close()
        fc["d%d" % d] = dict(diagonal_ms=t_diag * 1e3, full_ms=t_full * 1e3, ratio=t_full / t_diag, timed="in turn, median of 7")
    out["full_cov"] = fc
    d, n = 16, 8192
    vl, _k = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng = M.Engine(d, n, pl=1.0)
    t_small = time_job(eng, vl, pinit_for(d, n, 0), nsamp, nburn, reps=11)
    st = job_stats(eng)
    # the same two shapes in turn, on the same clock: the 65 536-chain job and this one
    same_clock = None
    try:
        big = M.Engine(d, headline_n, pl=1.0)
        big.stage_pinit(pinit_for(d, headline_n, 0))
        t_big_turn, t_small_turn = time_jobs_in_turn([lambda: big.run(nsamp, nburn, None, vl), lambda: eng.run(nsamp, nburn, None, vl)], reps=9)
        big.close()
        same_clock = dict(headline_shape_ms=t_big_turn * 1e3, ms_per_job=t_small_turn * 1e3, speedup=t_big_turn / t_small_turn,
                          timed="the two jobs in turn, median of 9")
    except Exception as ex:  # noqa: BLE001
        same_clock = dict(error=repr(ex))
    # the same jobs queued back to back (MCX_OPT_ASYNC_RUN: mcx_run returns once the run is queued, at most two in flight;
    # the clock stops after mcx_synchronize): a job's launch and completion latency under the job before it.  The
    # synchronous figure above is the one a single strong-scaled job sees; this is what a stream of them gets.
    piped = None
    try:
        eng.set_option(E.OPT_ASYNC_RUN, 1)
        for _ in range(4):
            eng.run(nsamp, nburn, None, vl)
        eng.synchronize()
        kq, batches = 60, []
        for _ in range(3):  # (the median of three batches: one stall of the box -- 13 ms were seen once -- is not the figure)
            t0 = time.perf_counter()
            for _ in range(kq):
                eng.run(nsamp, nburn, None, vl)
            eng.synchronize()
            batches.append((time.perf_counter() - t0) / kq)
        t_piped = sorted(batches)[1]
        eng.set_option(E.OPT_ASYNC_RUN, 0)
        piped = dict(ms_per_job=t_piped * 1e3, jobs=kq, batches_ms=[round(b * 1e3, 4) for b in batches],
                     speedup_vs_headline_job=headline_ms / (t_piped * 1e3), meet_timeouts_total=eng.counters["meet_timeouts_total"])
    except Exception as ex:  # noqa: BLE001
        piped = dict(error=repr(ex))
    eng.close()
    out["strong_proxy"] = dict(chains=n, ms_per_job=t_small * 1e3, value=n * (nburn + nsamp) / t_small, unit="chain-steps/s",
                               headline_chains=headline_n, headline_ms_per_job=headline_ms,
                               speedup_vs_headline_job=headline_ms / (t_small * 1e3), stats=st,
                               in_turn_with_the_headline_shape=same_clock, queued_back_to_back=piped)
    d, n = 16, 65536
    eng = M.Engine(d, n, pl=1.0)
    eng.run(20, 100, pinit_for(d, n, 0), vl)
    acc = eng.counters["naccept_main"] / float(n * 20)
    eng.close()
    out["accept_rate_reference_shape"] = dict(value=acc, reference=0.0468, workload="Rosenbrock1(16) x 65 536 chains, nburn 100 + nsamp 20, pl = 1")
    return out


# ---------------------------------------------------------------------------------------------
# one configuration on this rank's engine
# ---------------------------------------------------------------------------------------------
class Job:
    def __init__(self, M, E, cfg, n, rank, world, emit=True, max_segment=0):
        self.M, self.E, self.cfg, self.n = M, E, cfg, n
        self.eng = M.Engine(cfg["d"], n, nshards=world, shard=rank, pl=cfg["pl"])
        self.eng.set_option(E.OPT_SAMPLES, 1 if emit else 0)
        if max_segment > 0:
            self.eng.set_option(E.OPT_MAX_SEGMENT, max_segment)
        self.vl, self._keep = make_lik(M, cfg)
        self.p = pinit_for(cfg["d"], n, rank * n)
        self.eng.stage_pinit(self.p)  # inputs resident in HBM before any timed region

    def run(self, host_pinit=False):
        self.eng.run(self.cfg["nsamp"], self.cfg["nburn"], self.p if host_pinit else None, self.vl)

    def close(self):
        self.eng.close()


# ---------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process starts the N ranks itself
# ---------------------------------------------------------------------------------------------
EXIT_TOO_FEW_DEVICES = 66   # a rank found fewer GPUs than ranks (and no --one-device)
EXIT_RCCL_TIMEOUT = 75      # the RCCL communicator / its first gather did not come up within --rccl-timeout
EXIT_WATCHDOG = 124         # the parent saw no JSON line within --launch-timeout


def launch_ranks(n_ranks, argv, timeout_s):
    """`python bench.py --gpus N` called directly (WORLD_SIZE unset): start N fresh rank processes of this same script
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would), relay rank 0's one
    JSON line, return the exit status.  This parent never loads libmcx, torch or anything else that touches the GPU,
    and nothing that has touched the GPU re-execs: the ranks are children, started before any HIP call.
    A rank that dies takes the others with it; no line within `timeout_s` kills them all (exit 124).  Ranks that give
    up on RCCL (exit 75) are started ONCE more on the host-staged exchange, and the line then says why."""
    import signal
    import socket
    import threading

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        return p

    def attempt(extra, budget_s):
        port = free_port()
        procs, lines = [], []
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCX_BENCH_LAUNCHED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + extra, env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))

        def reader():
            for ln in procs[0].stdout:
                lines.append(ln)
        th = threading.Thread(target=reader, daemon=True)
        th.start()

        def kill_all():
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, signal.SIGTERM)
                    except OSError:
                        pass
            t_end = time.time() + 5.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except OSError:
                        pass
                    p.wait()
        t0 = time.time()
        rc = 0
        try:
            while True:
                codes = [p.poll() for p in procs]
                bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                if bad:
                    rc = bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]
                    for special in (EXIT_RCCL_TIMEOUT, EXIT_TOO_FEW_DEVICES):
                        if any(c == special for _r, c in bad):
                            rc = special
                    print("bench: rank %d exited with status %d; stopping the other ranks" % bad[0], file=sys.stderr)
                    kill_all()
                    break
                if all(c == 0 for c in codes):
                    break
                if time.time() - t0 > budget_s:
                    print("bench: no result after %.0f s (--launch-timeout): killing the %d ranks" % (budget_s, n_ranks), file=sys.stderr)
                    kill_all()
                    rc = EXIT_WATCHDOG
                    break
                time.sleep(0.05)
        except BaseException:
            kill_all()
            raise
        th.join(timeout=5.0)
        return rc, lines, time.time() - t0

    t_start = time.time()
    rc, lines, took = attempt([], timeout_s)
    if rc == EXIT_RCCL_TIMEOUT and "--exchange" not in argv:
        left = timeout_s - (time.time() - t_start)
        print("bench: the RCCL exchange did not come up; one more attempt on the host-staged all-gather (%.0f s left)" % left, file=sys.stderr)
        if left > 30:
            rc, lines, took = attempt(["--exchange", "staged", "--exchange-fallback-reason",
                                       "fallback because the RCCL communicator or its first all-gather did not complete in time"], left)
    if rc != 0:
        return rc
    good = []
    for ln in lines:
        try:
            o = json.loads(ln.decode("utf-8", "replace"))
        except ValueError:
            continue
        if isinstance(o, dict) and ("metric" in o or o.get("dry_launch")):
            good.append((ln, o))
    if len(good) != 1:
        print("bench: expected ONE JSON line from rank 0, got %d" % len(good), file=sys.stderr)
        return 1
    ln, o = good[0]
    if o.get("n_gpus") != n_ranks:
        print("bench: rank 0 reports n_gpus = %r, launched %d ranks" % (o.get("n_gpus"), n_ranks), file=sys.stderr)
        return 1
    c = o.get("config") or {}
    if "metric" in o and not (c.get("rccl_comm_ranks") == n_ranks or "fallback because" in str(c.get("exchange_backend"))
                              or "requested" in str(c.get("exchange_backend"))):
        print("bench: the line names neither an RCCL communicator of %d ranks nor the reason for another exchange: %r / %r"
              % (n_ranks, c.get("rccl_comm_ranks"), c.get("exchange_backend")), file=sys.stderr)
        return 1
    o["launcher"] = dict(kind="bench.py's own (WORLD_SIZE was unset)", ranks=n_ranks, wall_s=time.time() - t_start)
    if "summary" in o:  # keep `summary` the LAST key
        o["summary"] = o.pop("summary")
    sys.stdout.write(json.dumps(o) + "\n")
    sys.stdout.flush()
    return 0


def dry_launch():
    """--dry-launch: prove that the ranks this script was started as (by its own launcher or by torch.distributed.run)
    find each other: rendezvous on gloo, one all-gather of the ranks, one line from rank 0.  No GPU, no libmcx."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen = [rank]
    # the launcher's two failure paths, for tests/test_bench_launcher_cpu.py: a rank that never arrives, a rank that dies
    if os.environ.get("MCX_BENCH_DEBUG_HANG_RANK") == str(rank):
        time.sleep(3600)
    if os.environ.get("MCX_BENCH_DEBUG_DIE_RANK") == str(rank):
        sys.exit(7)
    if world > 1:
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, (rank, int(os.environ.get("LOCAL_RANK", "0")), os.getpid()))
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        assert int(t.item()) == world * (world + 1) // 2
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(dry_launch=True, n_gpus=world, ranks=seen,
                              launched_by="bench.py" if os.environ.get("MCX_BENCH_LAUNCHED") else "external launcher")), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--chains", type=int, default=0, help="override: chains per GPU (weak scaling, the default) or in total (--strong)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --chains (default: the configuration's count) is the total, split evenly")
    ap.add_argument("--dim", type=int, default=0, help="override the configuration's dimension")
    ap.add_argument("--nburn", type=int, default=-1)
    ap.add_argument("--nsamp", type=int, default=-1)
    ap.add_argument("--pl", type=float, default=-1.0)
    ap.add_argument("--no-samples", action="store_true", help="summary-only mode (not the default metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn rocprofv3 --pmc children; read profiles/ instead")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip end_to_end and the other configurations")
    ap.add_argument("--keep-pmc", default="", help="directory in which to keep the rocprofv3 output of the PMC passes")
    ap.add_argument("--max-segment", type=int, default=0, help="cap on steps per fused launch (0 = engine default)")
    ap.add_argument("--exchange", default="rccl", choices=("rccl", "staged"),
                    help="N > 1: rccl = the library's in-place ncclAllGather (default); staged = host-staged all-gather "
                         "over a gloo group (rehearsals on one GPU only)")
    ap.add_argument("--try-rccl", action="store_true", help="with --one-device: attempt the RCCL communicator anyway (RCCL refuses two "
                    "ranks on one GPU: rehearses the collective fallback to the staged exchange)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses device 0 (implies --exchange staged)")
    ap.add_argument("--cull", type=int, default=-1, choices=(-1, 0, 1), help="MCX_OPT_CULL of the measured job (Murray sweeps: exact "
                    "exclusion of far Gaussians): -1 auto (default), 0 off, 1 on")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true", help="only prove that the N ranks start and rendezvous (no GPU work)")
    ap.add_argument("--launch-timeout", type=float, default=540.0, help="--gpus N > 1 started directly: seconds the parent waits "
                    "for rank 0's line before it kills the ranks and exits 124")
    ap.add_argument("--rccl-timeout", type=float, default=90.0, help="seconds a rank waits for the RCCL communicator and its first "
                    "all-gather before it gives up (exit 75; bench.py's own launcher then retries on the staged exchange)")
    ap.add_argument("--extras-budget", type=float, default=240.0, help="N > 1: seconds after start beyond which no further extra "
                    "leg (reference schedule, Murray configuration, strong scaling) is begun")
    ap.add_argument("--exchange-fallback-reason", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        # called the way the driver calls it (`python3 bench.py --gpus 8 ...`, no launcher): be the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    if args.dry_launch:
        dry_launch()
        return

    cfg = dict(CONFIGS[args.config])
    if args.dim > 0:
        cfg["d"] = args.dim
    if args.nburn >= 0:
        cfg["nburn"] = args.nburn
    if args.nsamp >= 0:
        cfg["nsamp"] = args.nsamp
    if args.pl >= 0:
        cfg["pl"] = args.pl

    if args.pmc_child:
        pmc_child(cfg, args.chains or cfg["n"])
        return

    t_process_start = time.time()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1 and not args.one_device:
        # fewer GPUs than ranks is an error, never a quiet N = 1 (or N ranks on one device): every rank checks before
        # the rendezvous, so all of them leave with the same status
        from mcpar_amd import engine as _E
        try:
            ndev = _E.device_count()
        except Exception as ex:  # noqa: BLE001
            print("bench: rank %d: %s" % (rank, ex), file=sys.stderr)
            ndev = 0
        if ndev < world:
            print("bench: rank %d: --gpus %d but this node shows %d GPU(s) to libmcx (mcx_device_count); refusing to run "
                  "(rehearsals on one GPU: add --one-device)" % (rank, world, ndev), file=sys.stderr)
            sys.exit(EXIT_TOO_FEW_DEVICES)
    # stdout carries exactly ONE line, the JSON.  gloo announces its connections and RCCL prints its version banner
    # on the C-level stdout whenever a communicator is made (also the later ones: the Murray and strong-scaling
    # jobs), child profilers chatter too: file descriptor 1 points at stderr from here to the end of the run and
    # the line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world > 1:
        # torch.distributed is the launcher-side plumbing only (rendezvous, barrier, MAX over ranks, shipping the
        # RCCL unique id) on a gloo group; the data path is libmcx's own ncclAllGather
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        dist.init_process_group("gloo")
        dist.barrier()
        if args.one_device:
            local_rank = 0
            if not args.try_rccl:
                args.exchange = "staged"
    import numpy as np
    import mcpar_amd as M
    from mcpar_amd import engine as E

    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    lib = M.load()
    lib.mcx_set_device(local_rank)
    total = args.chains or cfg["n"]
    n = total // world if args.strong else total
    cfg["n"] = n
    d, nburn, nsamp = cfg["d"], cfg["nburn"], cfg["nsamp"]
    emit = not args.no_samples

    # ---- inter-shard exchange -------------------------------------------------------------------
    state = {"backend": None, "rccl_ranks": None, "rccl_bring_up_s": None}
    pci = None
    try:
        buf = C.create_string_buffer(64)
        if lib.mcx_device_pci_bus_id(buf, 64) == 0:
            pci = buf.value.decode()
    except Exception:  # noqa: BLE001
        pass
    pci_ids = [pci]
    if world > 1:  # which GPU every rank sits on (two ranks on one GPU cannot form an RCCL communicator)
        pci_ids = [None] * world
        dist.all_gather_object(pci_ids, pci)

    def bcast_bytes(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def new_rccl_id():
        """collective: rank 0 draws the ncclUniqueId, everybody gets it; None if RCCL is not usable everywhere"""
        ok = 1 if M.engine.rccl_available() else 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            return None
        uid = M.engine.rccl_unique_id() if rank == 0 else bytes(128)
        return bcast_bytes(uid)

    def staged_exchange(eng):
        host = {}

        def exchange(phase, ptr, slot, shard, nshards, st):
            if phase != E.XCHG_BEGIN:
                return 0
            h = host.setdefault("buf", np.empty(slot * nshards, np.float32))
            vp = C.c_void_p
            rc = lib.mcx_copy_to_host(vp(h.ctypes.data + shard * slot * 4), vp(ptr + shard * slot * 4), slot * 4, vp(st))
            if rc:
                return rc
            t = torch.from_numpy(h)
            dist.all_gather_into_tensor(t, t[shard * slot:(shard + 1) * slot].clone())
            for r in range(nshards):
                if r != shard:
                    rc = lib.mcx_copy_to_device(vp(ptr + r * slot * 4), vp(h.ctypes.data + r * slot * 4), slot * 4, vp(st))
                    if rc:
                        return rc
            return 0
        eng.set_exchange(exchange)

    def make_job(c, nn, emit_=True):
        if world == 1:
            return Job(M, E, c, nn, 0, 1, emit_, args.max_segment)
        uid = new_rccl_id() if args.exchange == "rccl" else None
        j = Job(M, E, c, nn, rank, world, emit_, args.max_segment)
        why = (args.exchange_fallback_reason or "requested") if args.exchange == "staged" else "fallback because RCCL is not loadable on every rank"
        if uid is not None:
            # communicator + one gather now: a broken fabric shows here, not inside the timed region.  Every rank
            # learns whether ALL ranks succeeded, so the fallback is taken by all of them or by none.
            # ncclCommInitRank / the first gather can hang on a broken fabric and cannot be cancelled: they run in a
            # thread (ctypes drops the GIL), and a rank that is still waiting after --rccl-timeout leaves with status 75
            import threading
            res = {"err": "", "t": None}

            def bring_up():
                t0 = time.time()
                try:
                    j.eng.rccl_init(uid)
                    # one real gather with known contents: every slot must arrive, whole, at every rank
                    if not j.eng.exchange_self_check():
                        res["err"] = "RCCL all-gather self-check failed: a slot did not arrive as sent"
                except Exception as ex:  # MCX_ERR_EXCHANGE with the RCCL error string
                    res["err"] = str(ex)
                res["t"] = time.time() - t0
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(args.rccl_timeout)
            if th.is_alive():
                print("bench: rank %d: the RCCL communicator / its first all-gather is not up after %.0f s (--rccl-timeout); "
                      "giving up (exit %d)" % (rank, args.rccl_timeout, EXIT_RCCL_TIMEOUT), file=sys.stderr)
                sys.stderr.flush()
                os._exit(EXIT_RCCL_TIMEOUT)
            err = res["err"]
            state["rccl_bring_up_s"] = res["t"]
            flag = torch.tensor([1 if err else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                try:
                    j.eng.rccl_destroy()
                except Exception:
                    pass
                uid = None
                why = "fallback because " + (err or "RCCL failed on another rank")
                print("bench: rank %d: RCCL exchange unavailable (%s); using the host-staged all-gather" % (rank, why), file=sys.stderr)
        if uid is None:
            staged_exchange(j.eng)
            state["backend"] = "gloo, host-staged (%s)" % why
            state["rccl_ranks"] = None
        else:
            state["backend"] = "rccl (libmcx in-place ncclAllGather on a side stream)"
            state["rccl_ranks"] = j.eng.rccl_info()[0]  # ncclCommCount of the communicator the gathers run on
        return j

    def sync():
        if world > 1:
            dist.barrier()  # run() is synchronous: it returns after this rank's stream has drained

    def timed(job, k, host_pinit=False):
        """k jobs bracketed by barriers, MAX over ranks"""
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            job.run(host_pinit)
        job.eng.synchronize()  # a sharded run's last all-gather may still be in flight (MCX_OPT_ASYNC_TAIL): inside the clock
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    skipped = []

    def within_budget(what):
        """N > 1: the extra legs are begun only while the slowest rank is inside --extras-budget (collective: all ranks
        take the same decision); what is left out is named in config.extras_skipped"""
        if world == 1:
            return True
        tt = torch.tensor([time.time() - t_process_start], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if float(tt.item()) < args.extras_budget:
            return True
        skipped.append(what)
        return False

    job = make_job(cfg, n, emit)
    eng = job.eng
    if args.cull != -1:
        eng.set_option(E.OPT_CULL, args.cull)
    for _ in range(args.warmup):
        job.run()
    dt = timed(job, args.steps)
    cnt = eng.counters
    chain_steps = float(world) * n * (nburn + nsamp) * args.steps
    value = chain_steps / dt
    value_host_pinit = chain_steps / timed(job, args.steps, host_pinit=True)  # PCIe-inclusive input, never `value`

    # N > 1: the same job with the reference's own exchange schedule (an all-gather at every sync point,
    # src/mcpar.cc:127-140, overlapped with the next segment) next to the default, which gathers only the
    # snapshots a Murray step or the end of the run reads (bit-identical results: tests/test_gpu_multishard.py)
    ref_sched = None
    if world > 1 and not args.no_extras and within_budget("reference_schedule"):
        eng.set_option(E.OPT_EAGER_EXCHANGE, 1)
        job.run()
        ke = max(1, args.steps // 2)
        de = timed(job, ke)
        ref_sched = dict(value=float(world) * n * (nburn + nsamp) * ke / de, unit="chain-steps/s", steps=ke,
                         ms_per_step=de / ke * 1e3, stats=job_stats(eng))  # MCX_OPT_EAGER_EXCHANGE = 1: a gather per sync point
        eng.set_option(E.OPT_EAGER_EXCHANGE, 0)

    # ---- live timing of the dominant kernels: HIP events on the engine's stream (MCX_OPT_PROFILE) --------
    def profiled(j):
        j.eng.set_option(E.OPT_PROFILE, 1)
        base = j.eng.profile
        j.run()
        pr = j.eng.profile
        j.eng.set_option(E.OPT_PROFILE, 0)
        return {k: {f: pr[k][f] - base[k][f] for f in ("ms", "launches", "chain_steps")} for k in pr}, j.eng.counters

    prof, pcnt = profiled(job)  # every rank runs it (it contains the same collectives)
    sync()

    def murray_block(c, nn, pr, cn):
        """k_remote_sweep in ALGORITHMIC flops (n_act N (3d+4) per pass; DESIGN.md 6 "the bench line")"""
        sw = pr["remote_sweep"]
        if sw["launches"] <= 0 or sw["ms"] <= 0:
            return None
        dd = c["d"]
        flops = sw["chain_steps"] * (3 * dd + 4)
        ach = flops / (sw["ms"] * 1e-3)
        kept = (cn.get("remote_pairs_evaluated", 0) / float(cn["remote_pairs"]) if cn.get("remote_pairs") else None)
        out = dict(bound="valu", kernel="k_remote_sweep*", achieved=ach / 1e12, peak=FP32_VALU_PEAK / 1e12,
                   unit="TFLOP/s", frac=ach / FP32_VALU_PEAK, pairs=sw["chain_steps"], flop_per_pair=3 * dd + 4,
                   launches=sw["launches"], total_ms=sw["ms"], remote_steps=cn["remote_steps"], passes=cn["remote_passes"],
                   whole_genremote_ms=pr["remote"]["ms"], pairs_evaluated=cn.get("remote_pairs_evaluated"),
                   pairs_evaluated_frac=kept,
                   pairs_evaluated_note="approximate: the groups of 128 chains the masks are made for depend on a counting sort's "
                                        "atomics, so the count of pairs left varies by a few in ten thousand from run to run; "
                                        "results and pass counts do not")
        if kept is not None:
            # `achieved` counts the algorithm's pairs, skipped or not (with the screens it passes the vector peak: the
            # point of them); this is the rate on the pairs the sweeps actually swept
            out["achieved_on_evaluated_pairs"] = ach * kept / 1e12
            out["frac_on_evaluated_pairs"] = ach * kept / FP32_VALU_PEAK
        sc = pr.get("remote_screen")
        if sc and sc["launches"] > 0 and sc["ms"] > 0:
            # the per-pair screen (mcx_screen.hpp): one bf16 product row of K = 2 np + 16 per pair on the matrix cores
            kk = 2 * dd + 16
            sf = sc["chain_steps"] * 2.0 * kk / (sc["ms"] * 1e-3)
            out["screen"] = dict(bound="mfma", kernel="k_screen_gemm<%d>" % dd, achieved=sf / 1e12, peak=BF16_MFMA_PEAK / 1e12,
                                 unit="TFLOP/s", frac=sf / BF16_MFMA_PEAK, pairs=sc["chain_steps"], flop_per_pair=2 * kk,
                                 launches=sc["launches"], total_ms=sc["ms"])
        return out

    roofline = murray = cpu = end_to_end = claims = host_cb = device_vl = None
    others = {}
    if rank == 0:
        fm, fb, rs = prof["fused_main"], prof["fused_burn"], prof["run_small"]
        if rs["launches"] > 0 and rs["ms"] > 0:
            # small-n mode: burn-in and main-loop steps run in ONE launch of k_run_small (mcx_persist.hpp)
            t_launch = rs["ms"] * 1e-3 / rs["launches"]
            abytes = float(n) * (nburn * alg_bytes_per_chain_step(d, False, False) + nsamp * alg_bytes_per_chain_step(d, True, emit))
            frac_local = rs["chain_steps"] / float(n * (nburn + nsamp))  # launches cover this share of the job's local steps
            alg = abytes * frac_local / (rs["ms"] * 1e-3)
            roofline = dict(bound="valu", kernel=None, kernel_family="k_run_small", avg_launch_ms=t_launch * 1e3, launches=rs["launches"],
                            chain_steps_per_launch=rs["chain_steps"] / rs["launches"],
                            hbm_alg=dict(achieved=alg / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=alg / HBM_PEAK))
        elif fm["launches"] > 0 and fm["ms"] > 0:
            bpc = alg_bytes_per_chain_step(d, True, emit)
            t_launch = fm["ms"] * 1e-3 / fm["launches"]
            alg = fm["chain_steps"] * bpc / (fm["ms"] * 1e-3)
            roofline = dict(bound="valu", kernel=None, kernel_family="k_fused_fast", avg_launch_ms=t_launch * 1e3, launches=fm["launches"],
                            chain_steps_per_launch=fm["chain_steps"] / fm["launches"],
                            hbm_alg=dict(achieved=alg / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=alg / HBM_PEAK, bytes_per_chain_step=bpc),
                            burn_kernel=dict(bytes_per_chain_step=alg_bytes_per_chain_step(d, False, False),
                                             avg_launch_ms=fb["ms"] / max(fb["launches"], 1), launches=fb["launches"],
                                             hbm_alg_GBps=(fb["chain_steps"] * alg_bytes_per_chain_step(d, False, False)
                                                           / max(fb["ms"] * 1e-3, 1e-12)) / 1e9))
        if cfg["pl"] < 1.0:
            murray = murray_block(cfg, n, prof, pcnt)
    headline_stats = job_stats(eng)

    # ---- N > 1: a Murray (pl < 1) configuration over the same ranks, first-class next to the R-local number ----
    murray_multi = None
    if world > 1 and cfg["pl"] >= 1.0 and not args.no_extras and within_budget("murray"):
        job.close()
        c5 = dict(CONFIGS["c5"])
        j5 = make_job(c5, c5["n"], True)
        j5.run()
        k5 = max(1, min(3, args.steps))
        d5 = timed(j5, k5)
        st5 = job_stats(j5.eng)
        p5, c5cnt = profiled(j5)
        sync()
        murray_multi = dict(workload=workload_text(c5, c5["n"]), value=float(world) * c5["n"] * (c5["nburn"] + c5["nsamp"]) * k5 / d5,
                            unit="chain-steps/s", steps=k5, ms_per_step=d5 / k5 * 1e3, remote_steps=c5cnt["remote_steps"],
                            passes=c5cnt["remote_passes"], stats=st5,
                            sweep=murray_block(c5, c5["n"], p5, c5cnt) if rank == 0 else None)
        j5.close()
        job = None

    # ---- N > 1: the same total job split over the ranks (strong scaling; `value` above is weak scaling) ----------
    strong_multi = None
    if world > 1 and not args.strong and not args.no_extras and total % world == 0 and within_budget("strong_scaling"):
        if job is not None:
            job.close()
            job = None
        cs = dict(cfg)
        ns_ = total // world
        cs["n"] = ns_
        js = make_job(cs, ns_, emit)
        strong_multi = dict(workload=workload_text(cs, ns_), chains_total=total, chains_per_gpu=ns_, unit="chain-steps/s")
        # default: a launch with tuner meetings starts behind the engine's own in-flight gather; then with the next run's
        # burn-in under it (MCX_OPT_MEET_UNDER_GATHER = 1; the meetings' timeout is the net: meet_timeouts says if it was needed)
        for key, under in (("behind_the_gather", 0), ("under_the_gather", 1)):
            js.eng.set_option(E.OPT_MEET_UNDER_GATHER, under)
            for _ in range(3):
                js.run()
            ks = max(10, args.steps)
            ds_ = timed(js, ks)
            strong_multi[key] = dict(value=float(total) * (nburn + nsamp) * ks / ds_, steps=ks, ms_per_step=ds_ / ks * 1e3,
                                     stats=job_stats(js.eng))
        strong_multi["value"] = strong_multi["behind_the_gather"]["value"]
        strong_multi["ms_per_step"] = strong_multi["behind_the_gather"]["ms_per_step"]
        js.close()

    # ---- N = 1 extras: end to end, other configurations, counters, CPU baseline ---------------------------
    if world == 1 and rank == 0:
        if emit and not args.no_extras:
            # the reference's product is the samples on the host (src/mcout.cc:30-94): the same job with every row
            # streamed out through the engine's sample sink (mcx_set_sink: device ring -> interleave -> pinned host
            # memory on a copy stream, overlapped with the steps), and -- for comparison -- the older way, the whole
            # run kept in HBM and copied out afterwards (mcx_samples_copy)
            seen = [0]

            def consumer(first, nsteps, rows):
                seen[0] += rows.shape[0]
                return 0
            blk = max(1, min(25, nsamp))
            eng.set_sink(consumer, blk)
            job.run()  # warm: pinned staging buffers are allocated here
            ke = max(1, min(3, args.steps))
            seen[0] = 0
            t0 = time.perf_counter()
            for _ in range(ke):
                job.run()
            de = (time.perf_counter() - t0) / ke
            eng.set_sink(None, 0)
            nbytes = nsamp * n * (d + 1) * 4
            assert seen[0] == ke * nsamp * n, "sink delivered %d rows, expected %d" % (seen[0], ke * nsamp * n)
            rows = np.empty((nsamp * n, d + 1), np.float32)
            job.run(); eng.samples_into(rows)  # warm (page-faults the destination once)
            t0 = time.perf_counter()
            for _ in range(ke):
                job.run()
                eng.samples_into(rows)
            dc = (time.perf_counter() - t0) / ke
            del rows
            # the reference's product is TEXT (src/mcout.cc:41-45, 65-87 % of its wall time): the last steps of the run as
            # the bytes MCout::output prints for them, formatted on the GPU (mcx_samples_text) and copied to the host
            text = None
            try:
                ts = min(25, nsamp)
                need = eng.samples_text_into(nsamp - ts, ts, None)
                tbuf = np.empty(need, np.uint8)
                eng.samples_text_into(nsamp - ts, ts, tbuf)  # warm (page-faults the destination once)
                t0 = time.perf_counter()
                got = eng.samples_text_into(nsamp - ts, ts, tbuf)
                dtx = time.perf_counter() - t0
                text = dict(steps=ts, numbers=ts * n * (d + 1), bytes=int(got), ms=dtx * 1e3, GBps=got / dtx / 1e9,
                            numbers_per_s=ts * n * (d + 1) / dtx, first_row=bytes(tbuf[:min(got, 60)]).decode("ascii", "replace"))
                del tbuf
            except Exception as ex:  # noqa: BLE001
                text = dict(error=repr(ex))
            # and the whole job with every row delivered AS TEXT (mcx_set_text_sink): what a driver that prints its samples
            # would hand to write(2)
            try:
                tbytes = [0]

                def tsink(first, nsteps, view):
                    tbytes[0] += len(view)
                    return 0
                eng.set_text_sink(tsink, blk)
                job.run()
                tbytes[0] = 0
                t0 = time.perf_counter()
                job.run()
                dts = time.perf_counter() - t0
                eng.set_text_sink(None, 0)
                if isinstance(text, dict):
                    text["whole_job_through_the_text_sink"] = dict(ms_per_step=dts * 1e3, bytes=int(tbytes[0]), GBps=tbytes[0] / dts / 1e9,
                                                                   value=n * (nburn + nsamp) / dts, unit="chain-steps/s")
            except Exception as ex:  # noqa: BLE001
                if isinstance(text, dict):
                    text["whole_job_through_the_text_sink"] = dict(error=repr(ex))
            end_to_end = dict(value=n * (nburn + nsamp) / de, unit="chain-steps/s", ms_per_step=de * 1e3, steps=ke,
                              host_bytes_per_job=int(nbytes), host_GBps=nbytes / de / 1e9, sink_block_steps=blk,
                              copy_after_the_run=dict(value=n * (nburn + nsamp) / dc, ms_per_step=dc * 1e3), text=text)
        job.close()
        job = None
        if not args.no_extras:
            for name in sorted(CONFIGS):
                if name == args.config:
                    continue
                c = dict(CONFIGS[name])
                j = Job(M, E, c, c["n"], 0, 1, True, args.max_segment)
                j.run()
                j.run()
                # an R-murray job draws its number of remote steps (binomial, 9 +- 3 of the 90 eligible steps) and of
                # rejection passes: a mean over 3 jobs moved by +-15 % from run to run of this script
                k = 10 if c["pl"] < 1.0 else 5
                t0 = time.perf_counter()
                for _ in range(k):
                    j.run()
                dd = (time.perf_counter() - t0) / k
                st = job_stats(j.eng)
                pr, cn = profiled(j)
                o = dict(workload=workload_text(c, c["n"]), value=c["n"] * (c["nburn"] + c["nsamp"]) / dd, unit="chain-steps/s",
                         ms_per_step=dd * 1e3, steps=k, accept_rate_main=cn["naccept_main"] / float(c["n"] * c["nsamp"]), stats=st)
                if c["pl"] < 1.0:
                    o["remote_steps"], o["passes"] = cn["remote_steps"], cn["remote_passes"]
                    o["sweep"] = murray_block(c, c["n"], pr, cn)
                others[name] = o
                j.close()
        # hardware counters of the dominant kernel, live
        if roofline is not None:
            try:  # the HBM figures also against a device-copy bandwidth measured on this box
                g = C.c_double(0.0)
                if lib.mcx_debug_copy_bandwidth(C.c_size_t(1 << 30), 10, C.byref(g)) == 0 and g.value > 0:
                    roofline["device_copy_GBps"] = dict(value=g.value, frac_of_nominal=g.value * 1e9 / HBM_PEAK)
            except Exception as ex:  # noqa: BLE001
                roofline["device_copy_GBps"] = dict(value=None, error=repr(ex))
            pmc = pdur = None
            note = "--no-pmc"
            if not args.no_pmc:
                pmc, pdur, note = collect_pmc(args.config, n, keep_dir=args.keep_pmc or None,
                                              extra_passes=(PMC_PASS_MFMA,) if cfg["pl"] < 1.0 else ())
            # the kernel that RAN: the one of the family the child job spent the most time in, under the name its kernel
            # trace gives it (k_fused_fastb<LPC2, BPL, ...> when the engine chose several blocks per lane, ...)
            dom = dominant_kernel(pmc, pdur, roofline["kernel_family"])
            kk = dom[0] if dom else None
            if kk:
                roofline["kernel"] = kk.split("(")[0].replace("void mcx::", "")
                roofline["kernel_trace"] = dict(avg_ns_under_the_profiler=dom[1], launches_per_job=dom[2])
            else:
                roofline["kernel"] = roofline["kernel_family"] + "<?> (no kernel trace: %s)" % note
            t_launch = roofline["avg_launch_ms"] * 1e-3
            small = roofline["kernel_family"] == "k_run_small"
            if kk and all(c in pmc[kk] for c in ("FETCH_SIZE", "WRITE_SIZE")):
                traffic = (2.0 * pmc[kk]["FETCH_SIZE"] + pmc[kk]["WRITE_SIZE"]) * 1024.0
                roofline["hbm_measured"] = dict(
                    traffic=traffic, achieved=traffic / t_launch / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                    frac=traffic / t_launch / HBM_PEAK, source="live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes",
                    frac_of_measured_copy_bandwidth=(traffic / t_launch / 1e9 / roofline["device_copy_GBps"]["value"]
                                                     if (roofline.get("device_copy_GBps") or {}).get("value") else None),
                    unavoidable_bytes=fm_unavoidable(d, n, roofline["chain_steps_per_launch"] * (nsamp / float(nburn + nsamp) if small else 1.0), emit))
                roofline["traffic"] = traffic
            if kk and all(c in pmc[kk] for c in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")):
                busy, insts, gui = pmc[kk]["SQ_ACTIVE_INST_VALU"], pmc[kk]["SQ_INSTS_VALU"], pmc[kk]["GRBM_GUI_ACTIVE"]
                cyc = gui / 8.0
                dur_ns = ((pdur.get(kk) or {}).get("GRBM_GUI_ACTIVE") or (None,))[0]
                waves = (n * lpc_for(d) + 63) // 64
                steps_per_launch = roofline["chain_steps_per_launch"] / n
                if small:
                    waves = 16 * min(waves, 256)  # every wavefront of the grid: owners, recorders, generators
                v = dict(SQ_INSTS_VALU=insts, SQ_ACTIVE_INST_VALU=busy, GRBM_GUI_ACTIVE=gui,
                         valu_instructions_per_wave_step=insts / (waves * steps_per_launch),
                         kernel_clock_GHz=(cyc / dur_ns) if dur_ns else None,
                         instructions_per_4_cycles_per_simd=4.0 * insts / (N_SIMD * cyc), source="live rocprofv3 --pmc child passes")
                need = ("SQ_INSTS_VALU_FLOPS_FP32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_FMA_F32",
                        "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32")
                if all(c in pmc[kk] for c in need):
                    c_ = pmc[kk]
                    fma, add, mul = c_["SQ_INSTS_VALU_FMA_F32"], c_["SQ_INSTS_VALU_ADD_F32"], c_["SQ_INSTS_VALU_MUL_F32"]
                    flops, trans, i64 = c_["SQ_INSTS_VALU_FLOPS_FP32"], c_["SQ_INSTS_VALU_TRANS_F32"], c_["SQ_INSTS_VALU_INT64"]
                    base = 2.0 * fma + add + mul  # flops if none of the f32 instructions were packed
                    packed = min(max((flops - base) / max(base, 1.0), 0.0), 1.0) * (fma + add + mul)  # v_pk_*_f32 count twice in FLOPS
                    plain = max(insts - packed - i64 - trans, 0.0)
                    issue = 2.0 * plain + 4.0 * (packed + i64) + 8.0 * trans  # wave64 on a SIMD-32: MI355X_MICROARCH.md
                    v.update(SQ_INSTS_VALU_FLOPS_FP32=flops, SQ_INSTS_VALU_FMA_F32=fma, SQ_INSTS_VALU_ADD_F32=add,
                             SQ_INSTS_VALU_MUL_F32=mul, SQ_INSTS_VALU_INT64=i64, SQ_INSTS_VALU_TRANS_F32=trans,
                             SQ_INSTS_VALU_IOPS=c_.get("SQ_INSTS_VALU_IOPS"), packed_f32_instructions_est=packed,
                             plain_instructions_est=plain, issue_cycles=issue,
                             frac_at_measured_issue_intervals=(2.7 * plain + 4.7 * (packed + i64) + 8.2 * trans) / (N_SIMD * cyc),
                             fp32_TFLOPs=(flops * 64.0 / t_launch / 1e12),
                             fp32_frac_of_vector_peak=(flops * 64.0 / t_launch) / FP32_VALU_PEAK)
                    roofline.update(achieved=issue / cyc, peak=float(N_SIMD), unit="SIMDs' worth of VALU issue slots in use (of 1024)",
                                    frac=issue / (N_SIMD * cyc))
                else:
                    roofline.update(achieved=2.0 * insts / cyc, peak=float(N_SIMD), unit="VALU instructions per 2 cycles (of 1024 SIMDs)",
                                    frac=2.0 * insts / (N_SIMD * cyc), frac_note="instruction classes not collected: every instruction at 2 cycles")
                roofline["valu"] = v
            if "frac" not in roofline or "traffic" not in roofline:
                # never another run's counters under this kernel's name: the figures stay empty and say why
                roofline.setdefault("frac", None)
                roofline.setdefault("achieved", None)
                roofline.setdefault("peak", float(N_SIMD))
                roofline.setdefault("unit", "SIMDs' worth of VALU issue slots in use (of 1024)")
                roofline.setdefault("traffic", None)
                roofline["counters_note"] = "no live counters for this kernel (%s)" % (note or "kernel not in the child's trace")
            if murray is not None and pmc:
                # the sweep kernels of the same child passes: how busy their VALU issue slots are
                vi = {}
                need = ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_FLOPS_FP32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT64",
                        "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32")
                for kk2, c_ in pmc.items():
                    if "k_remote_sweep" not in kk2 or not all(c in c_ for c in need):
                        continue
                    cyc2 = c_["GRBM_GUI_ACTIVE"] / 8.0
                    fma, add, mul = c_["SQ_INSTS_VALU_FMA_F32"], c_["SQ_INSTS_VALU_ADD_F32"], c_["SQ_INSTS_VALU_MUL_F32"]
                    base = 2.0 * fma + add + mul
                    packed = min(max((c_["SQ_INSTS_VALU_FLOPS_FP32"] - base) / max(base, 1.0), 0.0), 1.0) * (fma + add + mul)
                    plain = max(c_["SQ_INSTS_VALU"] - packed - c_["SQ_INSTS_VALU_INT64"] - c_["SQ_INSTS_VALU_TRANS_F32"], 0.0)
                    vi[kk2.split("(")[0].replace("void mcx::", "")] = dict(
                        SQ_INSTS_VALU=c_["SQ_INSTS_VALU"], packed_f32_instructions_est=packed, GRBM_GUI_ACTIVE=c_["GRBM_GUI_ACTIVE"],
                        frac=(2.0 * plain + 4.0 * packed) / (N_SIMD * cyc2),
                        frac_at_measured_issue_intervals=(2.7 * plain + 4.7 * packed) / (N_SIMD * cyc2))
                if vi:
                    murray["valu_issue"] = dict(kernels=vi)
                # the screen's kernel: cycles its SIMDs' matrix cores were busy (32 per v_mfma_f32_32x32x16_bf16) of the
                # cycles the kernel ran, from the same kind of child pass
                kg = find_kernel(pmc, "k_screen_gemm")
                if kg and murray.get("screen") and "SQ_VALU_MFMA_BUSY_CYCLES" in pmc[kg] and "GRBM_GUI_ACTIVE" in pmc[kg]:
                    cyc3 = pmc[kg]["GRBM_GUI_ACTIVE"] / 8.0
                    murray["screen"]["mfma_busy"] = dict(SQ_VALU_MFMA_BUSY_CYCLES=pmc[kg]["SQ_VALU_MFMA_BUSY_CYCLES"],
                                                         GRBM_GUI_ACTIVE=pmc[kg]["GRBM_GUI_ACTIVE"],
                                                         frac_of_simd_cycles=pmc[kg]["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc3),
                                                         note="mean over the dispatches of the measured child job (all pass sizes)")
        if not args.no_extras and args.config == "c3" and not args.chains and args.dim == 0:
            claims = claims_under_the_clock(M, E, dt / args.steps * 1e3, n, nburn, nsamp)
            host_cb = host_callback_leg(M, E, cfg, n)
            device_vl = device_vlfunc_leg(M, E, cfg, n, dt / args.steps * 1e3, emit)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg)

    if rank == 0:
        def pick(dct, *path):
            for k in path:
                if not isinstance(dct, dict) or dct.get(k) is None:
                    return None
                dct = dct[k]
            return dct
        sp = (claims or {}).get("strong_proxy")
        summary = {
            "value": value, "ms_per_job": dt / args.steps * 1e3, "n_gpus": world,
            "roofline_frac": pick(roofline, "frac"), "roofline_kernel": pick(roofline, "kernel"),
            "hbm_measured_frac": pick(roofline, "hbm_measured", "frac"),
            "meet_timeouts": headline_stats["meet_timeouts"], "exchange_wait_ms_per_run": headline_stats["exchange_wait_ms_per_run"],
            "other_configs_ms": {k: round(v["ms_per_step"], 4) for k, v in others.items()} or None,
            "other_configs_meet_timeouts": sum(v["stats"]["meet_timeouts"] for v in others.values()) if others else None,
            "c5_pairs_evaluated_frac": pick(others.get("c5"), "sweep", "pairs_evaluated_frac"),
            "c3_murray_pairs_evaluated_frac": pick(others.get("c3-murray"), "sweep", "pairs_evaluated_frac"),
            "c5_screen_mfma_frac": pick(others.get("c5"), "sweep", "screen", "frac") or pick(murray, "screen", "frac"),
            "c3_murray_screen_mfma_frac": pick(others.get("c3-murray"), "sweep", "screen", "frac"),
            "full_cov_ratio": {k: round(v["ratio"], 3) for k, v in ((claims or {}).get("full_cov") or {}).items() if isinstance(v, dict)} or None,
            # (the ratio's two terms: both jobs got faster in round 5, the diagonal one by more -- 16-D 2.25 / 2.74 ms and 32-D
            # 3.91 / 5.67 ms when the round began)
            "full_cov_ms_diagonal_full": {k: [round(v["diagonal_ms"], 3), round(v["full_ms"], 3)]
                                          for k, v in ((claims or {}).get("full_cov") or {}).items() if isinstance(v, dict)} or None,
            "strong_proxy_ms": pick(sp, "ms_per_job"), "strong_proxy_speedup": pick(sp, "speedup_vs_headline_job"),
            "strong_proxy_meet_timeouts": pick(sp, "stats", "meet_timeouts"),
            "strong_proxy_speedup_in_turn": pick(sp, "in_turn_with_the_headline_shape", "speedup"),
            "strong_proxy_queued_back_to_back_ms": pick(sp, "queued_back_to_back", "ms_per_job"),
            "strong_proxy_queued_back_to_back_speedup": pick(sp, "queued_back_to_back", "speedup_vs_headline_job"),
            "end_to_end_rows_ms": pick(end_to_end, "ms_per_step"),
            "end_to_end_text_ms": pick(end_to_end, "text", "whole_job_through_the_text_sink", "ms_per_step"),
            "host_callback_ms_per_step": pick(host_cb, "ms_per_step"), "host_callback_path_ms_per_step": pick(host_cb, "path_ms_per_step"),
            "device_vlfunc_source_fraction_of_builtin": (pick(device_vl, "source_block_form", "fraction_of_builtin"),
                                                         pick(device_vl, "source_whole_vector_form", "fraction_of_builtin")),
            "device_vlfunc_kernel_ms_per_step": pick(device_vl, "kernel", "ms_per_step"),
            "cpu_baseline": pick(cpu, "value"),
            "weak_reference_schedule_ms": pick(ref_sched, "ms_per_step"),
            "murray_multi_ms": pick(murray_multi, "ms_per_step"), "murray_multi_meet_timeouts": pick(murray_multi, "stats", "meet_timeouts"),
            "strong_multi_ms": pick(strong_multi, "ms_per_step"),
            "strong_multi_under_the_gather_ms": pick(strong_multi, "under_the_gather", "ms_per_step"),
            "strong_multi_meet_timeouts": (pick(strong_multi, "behind_the_gather", "stats", "meet_timeouts"),
                                           pick(strong_multi, "under_the_gather", "stats", "meet_timeouts")) if strong_multi else None,
            "strong_multi_exchange_wait_ms_per_run": pick(strong_multi, "behind_the_gather", "stats", "exchange_wait_ms_per_run"),
        }
        out = {
            "metric": "chain-steps/sec (all chains), 16-D Rosenbrock" if (cfg["lik"] == 1 and d == 16) else
                      "chain-steps/sec (all chains)",
            "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(cfg, n), "name": args.config,
                       "chains_per_gpu": n, "nparam": d, "nburn": nburn, "nsamp": nsamp, "pl": cfg["pl"],
                       "samples": "all kept in HBM" if emit else "none (summary only)",
                       "parallelism": ("chains sharded x%d (contiguous blocks, g = shard*n + j), in-place all-gather of the "
                                       "(mu, sig^2) slots" % world) if world > 1 else "single GPU",
                       "exchange_backend": state["backend"], "rccl_comm_ranks": state["rccl_ranks"],
                       "rccl_bring_up_s": state["rccl_bring_up_s"], "extras_skipped": skipped or None,
                       "rank0_wall_s": time.time() - t_process_start,
                       "pci_bus_ids": pci_ids,
                       "stats": headline_stats,
                       "reference_schedule": ref_sched,
                       "murray": murray_multi,
                       "strong_scaling": strong_multi,
                       "accept_rate_main": cnt["naccept_main"] / float(n * nsamp) if nsamp else None,
                       "remote_steps": cnt["remote_steps"], "remote_passes": cnt["remote_passes"],
                       "value_with_host_pinit": value_host_pinit,
                       "other_configs": others or None,
                       "full_cov": (claims or {}).get("full_cov"), "strong_proxy": sp,
                       "accept_rate_reference_shape": (claims or {}).get("accept_rate_reference_shape")},
            "end_to_end": end_to_end, "host_callback": host_cb, "device_vlfunc": device_vl, "murray_roofline": murray, "cpu_baseline": cpu, "roofline": roofline,
            "summary": summary,  # LAST: whoever keeps only the tail of this line still has every headline number
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if job is not None:
        job.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_callback_leg(M, E, cfg, n):
    """The path every user likelihood of the reference takes (src/vlfunc.hh:9-12 called at src/mcpar.cc:60,160; its R
    wrapper src/rfunc.cc:48-67): the configuration's shape with Rosenbrock1 as a HOST functor through MCX_VL_HOST -- the
    proposals of all chains copied to pinned host memory, the functor called once per step on x[n][d], its y[n] copied
    back.  The functor here is four numpy expressions; the time spent inside it is measured and reported apart from the
    path around it (propose / accept kernels, two PCIe copies, one stream sync per step)."""
    import numpy as np
    d = cfg["d"]
    inside = [0.0, 0]

    def rosen1(x):
        t0 = time.perf_counter()
        a, b = x[:, 0::2], x[:, 1::2]
        t1 = 1.0 - a
        t2 = b - a * a
        y = -(t1 * t1 + 100.0 * t2 * t2).sum(axis=1)
        inside[0] += time.perf_counter() - t0
        inside[1] += 1
        return y
    nburn, nsamp = 60, 40  # one tuner event, 40 main-loop steps: enough to time a per-step path
    vl, _keep = M.make_vlfunc(M.VL_HOST, d, host_fn=rosen1)
    eng = M.Engine(d, n, pl=1.0)
    eng.set_option(E.OPT_SAMPLES, 0)
    p = pinit_for(d, n, 0)
    eng.run(nsamp, nburn, p, vl)  # warm
    inside[0], inside[1] = 0.0, 0
    t0 = time.perf_counter()
    eng.run(nsamp, nburn, p, vl)
    dt = time.perf_counter() - t0
    calls = inside[1]
    eng.close()
    steps = nburn + nsamp
    bytes_per_step = n * d * 4 + n * 4
    path = (dt - inside[0]) / steps
    return dict(workload="Rosenbrock1(%d) as a host VLFunc (numpy) x %d chains, nburn %d + nsamp %d, no rows kept" % (d, n, nburn, nsamp),
                ms_per_step=dt / steps * 1e3, value=n * steps / dt, unit="chain-steps/s", functor_calls=calls,
                functor_ms_per_step=inside[0] / steps * 1e3, path_ms_per_step=path * 1e3,
                pcie_bytes_per_step=bytes_per_step, pcie_GBps_over_path_time=bytes_per_step / path / 1e9)


USER_KERNEL_TEMPLATE = r"""
// Rosenbrock1(%(d)d) as a user's own KERNEL (MCX_VL_DEVICE: the VLFunc contract of src/vlfunc.hh:9-12 on device memory),
// one thread per parameter set, sums in the engine's order (blocks of four, xor-butterfly over the blocks)
extern "C" __global__ void user_rosenbrock(int npset, const float *x, float *y)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npset) return;
  const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)j * %(d)d);
  float part[%(nb)d];
  for (int q = 0; q < %(nb)d; ++q) {
    const float4 v = p[q];
    const float a1 = 1.0f - v.x, a2 = __builtin_fmaf(-v.x, v.x, v.y);
    const float b1 = 1.0f - v.z, b2 = __builtin_fmaf(-v.z, v.z, v.w);
    part[q] = (0.0f + __builtin_fmaf(100.0f * a2, a2, a1 * a1)) + __builtin_fmaf(100.0f * b2, b2, b1 * b1);
  }
  for (int s = 1; s < %(nb)d; s <<= 1)
    for (int q = 0; q < %(nb)d; q += 2 * s)
      for (int r = 0; r < s; ++r) part[q + r] = part[q + r] + part[q + r + s];
  y[j] = 0.0f - part[0];
}
"""


def device_vlfunc_leg(M, E, cfg, n, headline_ms, emit):
    """A user's OWN likelihood on the GPU, both ways the C ABI offers (src/vlfunc.hh:9-12, called at src/mcpar.cc:60,160):
    `kernel` = MCX_VL_DEVICE, the user's separately compiled kernel between the engine's propose and accept kernels --
    three launches per step, the chain state round-tripping HBM every step, so SURVEY 8d's 392 (+68) B per chain-step
    apply literally; `source` = MCX_VL_SOURCE, the user's device functions compiled INTO the fused step kernels (hiprtc):
    the whole job, launch for launch like the built-in.  Both restate Rosenbrock1 so that the work is the headline's."""
    d, nburn, nsamp = cfg["d"], cfg["nburn"], cfg["nsamp"]
    out = {}
    p = pinit_for(d, n, 0)
    ex = os.path.join(ROOT, "mcpar_amd", "examples")
    for key, fname, par in (("source_block_form", "user_rosenbrock1_blocks.hip", None),
                            ("source_whole_vector_form", "user_rosenbrock1_whole.hip", [1.0])):
        try:
            t0 = time.perf_counter()
            vl, _keep = M.make_vlfunc(M.VL_SOURCE, d, params=par, source=open(os.path.join(ex, fname)).read())
            eng = M.Engine(d, n, pl=cfg["pl"])
            eng.set_option(E.OPT_SAMPLES, 1 if emit else 0)
            eng.stage_pinit(p)
            eng.run(nsamp, nburn, None, vl)  # includes the one-off hiprtc build of this text
            first = time.perf_counter() - t0
            t = time_job(eng, vl, p, nsamp, nburn, reps=7)
            st = job_stats(eng)
            acc = eng.counters["naccept_main"] / float(n * nsamp)
            eng.close()
            out[key] = dict(workload="C3's job with Rosenbrock1(%d) given as HIP source (%s), fused into the step kernels" % (d, fname),
                            ms_per_job=t * 1e3, value=n * (nburn + nsamp) / t, unit="chain-steps/s",
                            builtin_ms_per_job=headline_ms, fraction_of_builtin=headline_ms / (t * 1e3),
                            first_run_s_including_the_build=first, accept_rate_main=acc, stats=st)
        except Exception as ex_:  # noqa: BLE001
            out[key] = dict(error=repr(ex_))
    try:  # few chains (the strong-scaled per-GPU shape): the block form also gets the one-launch small-n kernel
        ns = 8192
        vl, _keep = M.make_vlfunc(M.VL_SOURCE, d, source=open(os.path.join(ex, "user_rosenbrock1_blocks.hip")).read())
        eng = M.Engine(d, ns, pl=cfg["pl"])
        eng.set_option(E.OPT_SAMPLES, 1 if emit else 0)
        t = time_job(eng, vl, pinit_for(d, ns, 0), nsamp, nburn, reps=11)
        st = job_stats(eng)
        eng.close()
        vb, _keepb = M.make_vlfunc(M.VL_ROSENBROCK1, d)
        eng = M.Engine(d, ns, pl=cfg["pl"])
        eng.set_option(E.OPT_SAMPLES, 1 if emit else 0)
        tb = time_job(eng, vb, pinit_for(d, ns, 0), nsamp, nburn, reps=11)
        eng.close()
        out["source_block_form_8192_chains"] = dict(workload="the same text, %d chains: the whole run in one launch of the small-n kernel" % ns,
                                                    ms_per_job=t * 1e3, value=ns * (nburn + nsamp) / t, unit="chain-steps/s",
                                                    builtin_ms_per_job=tb * 1e3, fraction_of_builtin=tb / t, stats=st)
    except Exception as ex_:  # noqa: BLE001
        out["source_block_form_8192_chains"] = dict(error=repr(ex_))
    try:
        fn = E.compile_user_kernel(USER_KERNEL_TEMPLATE % dict(d=d, nb=d // 4), "user_rosenbrock")
        vl, _keep = M.make_vlfunc(M.VL_DEVICE, d, device_fn=fn)
        kb, ks = 60, 40  # like host_callback: enough steps to time a per-step path
        eng = M.Engine(d, n, pl=1.0)
        eng.set_option(E.OPT_SAMPLES, 1 if emit else 0)
        t = time_job(eng, vl, p, ks, kb, reps=5)
        st = job_stats(eng)
        eng.close()
        abytes = float(n) * (kb * alg_bytes_per_chain_step(d, False, False) + ks * alg_bytes_per_chain_step(d, True, emit))
        out["kernel"] = dict(workload="Rosenbrock1(%d) as the user's own kernel (MCX_VL_DEVICE) x %d chains, nburn %d + nsamp %d: "
                                      "propose / user kernel / accept, three launches per step" % (d, n, kb, ks),
                             ms_per_step=t / (kb + ks) * 1e3, value=n * (kb + ks) / t, unit="chain-steps/s",
                             hbm_alg=dict(achieved=abytes / t / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=abytes / t / HBM_PEAK,
                                          note="SURVEY 8d bytes per chain-step; on this path the state does round-trip HBM every step"),
                             stats=st)
    except Exception as ex_:  # noqa: BLE001
        out["kernel"] = dict(error=repr(ex_))
    return out


def fm_unavoidable(d, n, chain_steps_per_launch, emit):
    """bytes one launch cannot avoid: the sample rows it emits + state and moments once in and once out"""
    rows = chain_steps_per_launch * 4 * (d + 1) if emit else 0
    return rows + n * (2 * 4 * (3 * d + 1))


if __name__ == "__main__":
    main()
