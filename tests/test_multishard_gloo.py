"""N > 1 path on CPU: two processes (torch.distributed, gloo, world_size 2), one shard each, run the
oracle engine with an all-gather exchange hook -- the same hook contract bench.py gives the HIP
engine with RCCL -- and must reproduce the in-process two-shard oracle run bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import oracle_lib as O

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d, n, nburn, nsamp, pl = 16, 48, 120, 60, 0.7
e = O.Engine(d, n, nshards=world, shard=rank, pl=pl)
calls = [0]

def exchange(musigall, slot, shard, nshards):
    # in-place all-gather of this shard's slot (the MPI_Allgather of src/mcpar.cc:127-140)
    t = torch.from_numpy(musigall)
    own = t[shard * slot:(shard + 1) * slot].clone()
    dist.all_gather_into_tensor(t, own)
    calls[0] += 1
    return 0

e.set_exchange(exchange)
vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
e.run(nsamp, nburn, O.default_pinit(d, n, g0=rank * n), vl)
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), state=e.state, mean=e.mean, var=e.var,
         mask=e.accept_mask, musigall=e.musigall, calls=calls[0], passes=e.remote_passes)
dist.barrier()
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_process_gloo_matches_in_process_shards(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d, n, nburn, nsamp, pl = 16, 48, 120, 60, 0.7
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    engs = [O.Engine(d, n, nshards=2, shard=s, pl=pl) for s in range(2)]
    O.run_all(engs, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(2)], vl)
    for s in range(2):
        got = np.load(tmp_path / ("rank%d.npz" % s))
        assert int(got["calls"]) == nsamp // 10
        assert int(got["passes"]) == engs[s].remote_passes and engs[s].remote_steps > 0
        assert np.array_equal(got["mask"], engs[s].accept_mask)
        for name in ("state", "mean", "var", "musigall"):
            assert np.array_equal(got[name].view(np.uint32), getattr(engs[s], name).view(np.uint32)), name


def test_shard_invariance_of_local_runs():
    """with pl = 1 chains never interact: 1 shard of 2n chains == 2 shards of n chains, except for
    the burn-in tuner, which the reference runs per rank; with the tuner disabled (armin = 0,
    armax = 2) the two layouts must agree bit for bit"""
    d, n = 8, 40
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    one = O.Engine(d, 2 * n, pl=1.0, armin=0.0, armax=2.0)
    one.run(30, 60, O.default_pinit(d, 2 * n), vl)
    two = [O.Engine(d, n, nshards=2, shard=s, pl=1.0, armin=0.0, armax=2.0) for s in range(2)]
    O.run_all(two, 30, 60, [O.default_pinit(d, n, g0=s * n) for s in range(2)], vl)
    both = np.concatenate([two[0].state, two[1].state])
    assert np.array_equal(one.state.view(np.uint32), both.view(np.uint32))
