"""Un-skippable evidence for the exchange hook the library ships: RCCL's two-rank tests need two GPUs
and skip on this pool, so the hook's own stream / event logic (mcx_exchange.hip:142-157: BEGIN on a side stream behind the
step stream, WAIT, the communicator bookkeeping, MCX_OPT_ASYNC_TAIL, destroy while a gather is pending) had never run with
a peer.  Here it does: engines in threads of one process on the one GPU, the collective itself supplied by
tests/cpp/rccl_stub.hip via MCX_RCCL_LIB, results against the oracle's in-process multi-shard run, bit for bit.
The exchange replaces MPI_Allgather(MPI_IN_PLACE, ..., musigall) of src/mcpar.cc:127-140."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rccl_stub") / "librccl_stub.so")
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-O2",
                           os.path.join(HERE, "cpp", "rccl_stub.hip"), "-o", so])
    return so


@pytest.mark.parametrize("nshards,pl,eager,async_tail,runs,persist", [
    (2, 0.8, 0, 1, 1, 1), (2, 0.8, 1, 1, 1, 1), (3, 0.7, 0, 1, 2, 1), (2, 1.0, 1, 1, 3, 1), (2, 0.85, 1, 0, 2, 0), (4, 0.9, 0, 1, 2, 0)],
    ids=["2-lazy", "2-eager", "3-lazy-2runs", "2-local-eager-3runs-tail-in-flight", "2-eager-no-async-tail-segments", "4-lazy-segments"])
def test_own_rccl_hook_with_a_peer_equals_oracle(stub, tmp_path, nshards, pl, eager, async_tail, runs, persist):
    d, n, nburn, nsamp = 16, 128, 120, 70
    env = dict(os.environ, MCX_RCCL_LIB=stub, RCCL_STUB_TIMEOUT_MS="30000")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_stub_worker.py"), str(nshards), str(d), str(n), str(nburn), str(nsamp),
                        str(pl), str(eager), str(async_tail), str(runs), str(persist), str(tmp_path)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode("utf-8", "replace")[-3000:]
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl) for s in range(nshards)]
    for e in eos:
        e.set_record(samples=True, mask=False)
    for _ in range(runs):
        O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    total_exchanges = 0
    for s in range(nshards):
        got = np.load(os.path.join(tmp_path, "shard%d.npz" % s))
        eo = eos[s]
        assert list(got["counters"][:3]) == [eo.remote_steps, eo.remote_passes, eo.naccept_main], s
        if eager:
            assert got["counters"][3] == (nsamp + 9) // 10
        total_exchanges = int(got["counters"][3])
        for name in ("state", "mean", "var", "musigall"):
            assert same_bits(got[name], getattr(eo, name)), (s, name)
        assert same_bits(got["samples"], eo.samples[-nsamp * n:]), s  # (the oracle keeps every run's rows, the engine the last run's)
        # the gathers went through the stub's ncclAllGather: 2 of the start-up self-check + every exchange of every run, per shard
        assert int(got["stub_calls"]) >= nshards * (2 + 1)
    assert total_exchanges >= 1


def test_bench_gives_up_on_a_communicator_that_never_comes_up_and_says_so(stub, tmp_path):
    """bench.py --gpus 2 started directly: the stub's ncclCommInitRank waits for a peer THREAD that never comes (the peer
    is another process): each rank leaves with status 75 after --rccl-timeout, bench.py's launcher starts the ranks once
    more on the host-staged exchange and the line carries the reason -- a line, not a hang, and never a quiet N = 1."""
    env = dict(os.environ, MCX_RCCL_LIB=stub, RCCL_STUB_TIMEOUT_MS="600000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--try-rccl", "--rccl-timeout", "8",
                        "--steps", "1", "--warmup", "1", "--no-extras", "--chains", "4096", "--nsamp", "100", "--nburn", "100"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    err = r.stderr.decode("utf-8", "replace")
    assert r.returncode == 0, err[-3000:]
    assert "giving up (exit 75)" in err and "one more attempt on the host-staged all-gather" in err
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["config"]["rccl_comm_ranks"] is None
    assert "fallback because" in o["config"]["exchange_backend"]
    assert o["launcher"]["ranks"] == 2
