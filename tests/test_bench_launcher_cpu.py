"""`python bench.py --gpus N` called directly, the way the round-end driver calls it (no torch.distributed.run, WORLD_SIZE
unset), must BE an N-rank run: bench.py starts the N rank processes itself before anything touches a GPU, relays rank 0's one
line, and fails loudly -- never a quiet N = 1 -- when a rank dies, hangs, or finds fewer GPUs than ranks.
The exchange the ranks would then run replaces MPI_Allgather of src/mcpar.cc:127-140 over mpisiz ranks (src/mcpar.cc:225)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return r, time.time() - t0


def test_two_ranks_rendezvous_without_a_launcher():
    r, _ = run(["--gpus", "2", "--one-device", "--dry-launch"])
    assert r.returncode == 0, r.stderr.decode()
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE line on stdout, whatever gloo prints
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["launched_by"] == "bench.py"
    assert sorted(x[0] for x in o["ranks"]) == [0, 1] and sorted(x[1] for x in o["ranks"]) == [0, 1]
    assert len({x[2] for x in o["ranks"]}) == 2  # two processes
    assert o["launcher"]["ranks"] == 2


def test_three_ranks_under_an_external_launcher_are_left_alone():
    """WORLD_SIZE set (torch.distributed.run's contract): bench.py is one rank and starts nothing"""
    r, _ = run(["--gpus", "3", "--dry-launch"], env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr.decode()
    o = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert o["n_gpus"] == 1 and o["launched_by"] == "external launcher" and "launcher" not in o


def test_fewer_gpus_than_ranks_is_an_error_not_a_quiet_single_rank():
    """this container has no GPU at all: every rank must refuse (status 66) and the parent must pass that on"""
    r, took = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"])
    assert r.returncode == 66, (r.returncode, r.stderr.decode())
    assert r.stdout.decode().strip() == ""
    assert "refusing to run" in r.stderr.decode()
    assert took < 60


def test_a_dead_rank_stops_the_others():
    r, took = run(["--gpus", "3", "--one-device", "--dry-launch"], env=dict(MCX_BENCH_DEBUG_DIE_RANK="2"))
    assert r.returncode == 7, (r.returncode, r.stderr.decode())
    assert r.stdout.decode().strip() == ""
    assert "rank 2 exited with status 7" in r.stderr.decode()
    assert took < 60  # the two live ranks were waiting at the rendezvous: killed, not waited for


def test_a_hung_rank_trips_the_watchdog():
    r, took = run(["--gpus", "2", "--one-device", "--dry-launch", "--launch-timeout", "6"], env=dict(MCX_BENCH_DEBUG_HANG_RANK="1"))
    assert r.returncode == 124, (r.returncode, r.stderr.decode())
    assert r.stdout.decode().strip() == ""
    assert "--launch-timeout" in r.stderr.decode()
    assert 5 < took < 40
