#!/usr/bin/env python3
"""bench.py -- chain-steps/sec of the MI355X-native Metropolis-Hastings engine on BASELINE.json's
metric config: 16-D Rosenbrock (Rosenbrock1(16): SURVEY fact 4), 65 536 chains per GPU, the R-local
job shape of SURVEY §8d (pl = 1.0, nburn = 500, nsamp = 1000, default tuner constants, seed 8675309,
pinit[g][i] = 0.5 sin(0.37 (g d + i))), every (chain, step) sample retained in HBM like the
reference's MCout.

A bench "step" is ONE complete MCPar::run-shaped job (burn-in with tuner + main loop with sample
emission + the inter-shard exchange every SYNCSTEP steps when N > 1) = n * (nburn + nsamp)
chain-steps per GPU.  Chains shard across GPUs (weak scaling: 65 536 chains per GPU).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver on this pool only supports dmabuf IPC (needed by RCCL across processes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md chip table


def alg_bytes_per_chain_step(d, main, emit):
    """state round trip 2*4*(3d+1) main / 2*4*(d+1) burn-in, + 4(d+1) per kept sample"""
    b = 8 * (3 * d + 1) if main else 8 * (d + 1)
    return b + (4 * (d + 1) if (main and emit) else 0)


def pinit_for(d, n, g0):
    import numpy as np
    g = np.arange(g0, g0 + n, dtype=np.float64)[:, None]
    i = np.arange(d, dtype=np.float64)[None, :]
    return (0.5 * np.sin(0.37 * (g * d + i))).astype(np.float32)


class CudaArrayView:
    """zero-copy torch view of engine-owned device memory (for the RCCL exchange)"""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = dict(shape=(nfloats,), typestr="<f4", data=(int(ptr), False),
                                             version=2, strides=None)


def cpu_baseline(d, n, nburn, nsamp, pl):
    """The CPU oracle (a port of the reference algorithm, oracle/mcx_oracle.c) on this box's host
    cores, on a bounded sample of the same workload.  Reported beside the GPU number; never part
    of the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the GPU box's CPU share for one GPU is 16 cores
    vl, _keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    p = pinit_for(d, n, 0)
    e = O.Engine(d, n, pl=pl, threads=cores)
    e.set_record(samples=True, mask=False)
    t0 = time.perf_counter()
    e.run(nsamp, nburn, p, vl)
    dt = time.perf_counter() - t0
    e.close()
    return dict(value=n * (nburn + nsamp) / dt, unit="chain-steps/s", cores=cores, kind="port",
                sample="%d chains x %d-D Rosenbrock1, %d burn-in + %d main steps, samples kept in host "
                       "memory, OpenMP over chains (%.1f s)" % (n, d, nburn, nsamp, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU (weak scaling, the default) or in total (--strong)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --chains is the total, split evenly over the GPUs")
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--nburn", type=int, default=500)
    ap.add_argument("--nsamp", type=int, default=1000)
    ap.add_argument("--pl", type=float, default=1.0)
    ap.add_argument("--no-samples", action="store_true", help="summary-only mode (not the default metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-segment", type=int, default=0, help="cap on steps per fused launch (0 = engine default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1:
        import torch  # noqa: F811  (before libmcx so that both use one HIP runtime)
        import torch.distributed as dist  # noqa: F811
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    import numpy as np
    import mcpar_amd as M
    from mcpar_amd import engine as E

    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    nsh = world
    M.load().mcx_set_device(local_rank)
    d, n, nburn, nsamp = args.dim, (args.chains // world if args.strong else args.chains), args.nburn, args.nsamp
    emit = not args.no_samples
    eng = M.Engine(d, n, nshards=nsh, shard=rank, pl=args.pl)
    eng.set_option(E.OPT_SAMPLES, 1 if emit else 0)
    if args.max_segment > 0:
        eng.set_option(E.OPT_MAX_SEGMENT, args.max_segment)
    vl, _keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    p = pinit_for(d, n, rank * n)

    state = {"backend": None}
    if world > 1:
        # engine kernels and the RCCL all-gather are ordered through torch's current stream
        stream = torch.cuda.current_stream()
        eng.set_option(E.OPT_STREAM, stream.cuda_stream)
        lib = M.load()
        # Safety net: a host-staged all-gather over a gloo group, used only if the zero-copy RCCL path
        # (in-place all_gather_into_tensor on a __cuda_array_interface__ view) does not work on this
        # stack.  The decision is taken once, collectively, on a probe buffer.
        gloo = dist.new_group(backend="gloo") if args.backend == "nccl" else None
        ok = 0 if os.environ.get("MCX_BENCH_FORCE_STAGED") else 1  # (rehearsal switch for the fallback)
        try:
            if not ok:
                raise RuntimeError("forced")
            probe = torch.zeros(world * 256, dtype=torch.float32, device="cuda")
            view = torch.as_tensor(CudaArrayView(probe.data_ptr(), probe.numel()), device="cuda")
            own = view[rank * 256:(rank + 1) * 256]
            own.fill_(float(rank + 1))
            dist.all_gather_into_tensor(view, own)
            torch.cuda.synchronize()
            got = probe.view(world, 256)[:, 0].cpu()
            ok = int(bool((got == torch.arange(1, world + 1, dtype=torch.float32)).all()))
        except Exception as ex:  # noqa: BLE001
            print("rank %d: zero-copy RCCL all-gather probe failed: %r" % (rank, ex), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=gloo)
        zero_copy = int(flag.item()) == 1
        state["backend"] = ("rccl" if args.backend == "nccl" else args.backend) if zero_copy else "gloo, host-staged (fallback)"

        def exchange(phase, ptr, slot, shard, nshards, st):
            if zero_copy:
                if phase == E.XCHG_BEGIN:
                    if "all" not in state:
                        state["all"] = torch.as_tensor(CudaArrayView(ptr, slot * nshards), device="cuda")
                        state["own"] = state["all"][shard * slot:(shard + 1) * slot]
                    state["work"] = dist.all_gather_into_tensor(state["all"], state["own"], async_op=True)
                else:
                    w = state.pop("work", None)
                    if w is not None:
                        w.wait()
                return 0
            if phase != E.XCHG_BEGIN:
                return 0
            host = state.setdefault("host", np.empty(slot * nshards, np.float32))
            vp = C.c_void_p
            rc = lib.mcx_copy_to_host(vp(host.ctypes.data + shard * slot * 4), vp(ptr + shard * slot * 4), slot * 4, vp(st))
            if rc:
                return rc
            t = torch.from_numpy(host)
            dist.all_gather_into_tensor(t, t[shard * slot:(shard + 1) * slot].clone(), group=gloo)
            for r in range(nshards):
                if r != shard:
                    rc = lib.mcx_copy_to_device(vp(ptr + r * slot * 4), vp(host.ctypes.data + r * slot * 4), slot * 4, vp(st))
                    if rc:
                        return rc
            return 0
        eng.set_exchange(exchange)

    def sync():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # inputs resident in HBM before the timed region: the initial chain state is staged once; the
    # host-pointer form run(pinit) is timed separately below (PCIe-inclusive, reported, never `value`)
    eng.stage_pinit(p)
    for _ in range(args.warmup):
        eng.run(nsamp, nburn, None, vl)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run(nsamp, nburn, None, vl)  # synchronous: returns after the stream has drained
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    cnt = eng.counters
    chain_steps = float(world) * n * (nburn + nsamp) * args.steps
    value = chain_steps / dt
    sync()
    th = time.perf_counter()
    for _ in range(args.steps):
        eng.run(nsamp, nburn, p, vl)  # pinit handed over as a host buffer each time
    sync()
    value_host_pinit = chain_steps / (time.perf_counter() - th)

    # N > 1: the same job with the reference's own exchange schedule (an all-gather at every sync
    # point, overlapped with compute) next to the default, which gathers only the snapshots a Murray
    # step or the end of the run will read (bit-identical results; tests/test_gpu_multishard.py)
    eager = None
    if world > 1:
        eng.set_option(E.OPT_EAGER_EXCHANGE, 1)
        eng.run(nsamp, nburn, None, vl)
        sync()
        t1 = time.perf_counter()
        ke = max(1, args.steps // 2)
        for _ in range(ke):
            eng.run(nsamp, nburn, None, vl)
        sync()
        de = time.perf_counter() - t1
        te = torch.tensor([de], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        eager = dict(value=float(world) * n * (nburn + nsamp) * ke / float(te.item()), unit="chain-steps/s",
                     steps=ke, exchanges_per_run=eng.counters["exchanges"])
        eng.set_option(E.OPT_EAGER_EXCHANGE, 0)

    # ---- roofline of the dominant kernel (fused main-loop steps), HIP events on the engine stream
    roofline = None
    cpu = None
    # every rank runs the profiled job (it contains the same collectives); rank 0 reports
    eng.set_option(E.OPT_PROFILE, 1)
    base = eng.profile
    eng.run(nsamp, nburn, None, vl)
    pr = eng.profile
    eng.set_option(E.OPT_PROFILE, 0)
    sync()
    if rank == 0:
        fm = {k: pr["fused_main"][k] - base["fused_main"][k] for k in ("ms", "launches", "chain_steps")}
        fb = {k: pr["fused_burn"][k] - base["fused_burn"][k] for k in ("ms", "launches", "chain_steps")}
        if fm["launches"] > 0 and fm["ms"] > 0:
            bpc = alg_bytes_per_chain_step(d, True, emit)
            achieved = fm["chain_steps"] * bpc / (fm["ms"] * 1e-3)
            # PMC numbers cannot be collected inside this process: they come from the committed rocprofv3
            # --pmc passes of this same command (profiles/, tools/collect_profiles.py)
            traffic = valu_busy = None
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(
                    "k_fused_steps_main_bytes_per_launch")
                valu_busy = json.load(open(os.path.join(ROOT, "profiles", "r01_fused_kernel_sq_counters.json"))).get(
                    "valu_busy_fraction")
            except Exception:
                pass
            roofline = dict(bound="hbm", kernel="k_fused_fast<LPC=%d,MAIN>" % max(1, (d + 3) // 4),
                            achieved=achieved / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                            frac=achieved / HBM_PEAK, traffic=traffic, valu_busy=valu_busy,
                            alg_bytes_per_chain_step=bpc,
                            avg_launch_ms=fm["ms"] / fm["launches"], launches=fm["launches"],
                            chain_steps_per_launch=fm["chain_steps"] / fm["launches"],
                            burn_kernel=dict(alg_bytes_per_chain_step=alg_bytes_per_chain_step(d, False, False),
                                             avg_launch_ms=fb["ms"] / max(fb["launches"], 1),
                                             achieved=(fb["chain_steps"] * alg_bytes_per_chain_step(d, False, False)
                                                       / max(fb["ms"] * 1e-3, 1e-12)) / 1e9))
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(d, n, 500, 500, args.pl)

    if rank == 0:
        out = {
            "metric": "chain-steps/sec (all chains), 16-D Rosenbrock",
            "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "mcpar-rosen2 (C3): Rosenbrock1(%d) x %d chains/GPU, R-local job "
                                   "(pl=%.2f, nburn=%d, nsamp=%d, sync=10), one bench step = one full run()"
                                   % (d, n, args.pl, nburn, nsamp),
                       "chains_per_gpu": n, "nparam": d, "nburn": nburn, "nsamp": nsamp,
                       "samples": "all kept in HBM" if emit else "none (summary only)",
                       "parallelism": "chains sharded x%d, RCCL all-gather of the (mu, sig^2) snapshots that a Murray "
                                      "step or the end of the run reads" % world if world > 1 else "single GPU",
                       "exchange_backend": state["backend"],
                       "eager_exchange": eager,
                       "accept_rate_main": cnt["naccept_main"] / float(n * nsamp),
                       "value_with_host_pinit": value_host_pinit,
                       "exchanges_per_run": cnt["exchanges"], "kernel_launches_per_run": cnt["kernel_launches"]},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
