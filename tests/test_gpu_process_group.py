"""The HIP engine's exchange hook under a real process group: two processes (torch.distributed, gloo, launched the way
bench.py is for N > 1), one shard each -- both on the box's one GPU, so the (mu, sig^2) slots go through the
host-staged all-gather bench.py falls back to when RCCL refuses two ranks on one device -- against the in-process
two-shard oracle run.  The gather replaces MPI_Allgather(MPI_IN_PLACE, ..., musigall) of src/mcpar.cc:127-140."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes as C, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import mcpar_amd as M
from mcpar_amd import engine as E
import oracle_lib as O

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d, n, nburn, nsamp, pl, eager = %(d)d, %(n)d, %(nburn)d, %(nsamp)d, %(pl)r, %(eager)d
lib = M.load()
lib.mcx_set_device(0)
eng = M.Engine(d, n, nshards=world, shard=rank, pl=pl)
eng.set_option(E.OPT_EAGER_EXCHANGE, eager)
host = np.empty(2 * n * d * world, np.float32)
calls = [0]

def exchange(phase, ptr, slot, shard, nshards, st):
    if phase != E.XCHG_BEGIN:
        return 0
    vp = C.c_void_p
    rc = lib.mcx_copy_to_host(vp(host.ctypes.data + shard * slot * 4), vp(ptr + shard * slot * 4), slot * 4, vp(st))
    if rc:
        return rc
    t = torch.from_numpy(host)
    dist.all_gather_into_tensor(t, t[shard * slot:(shard + 1) * slot].clone())
    for r in range(nshards):
        if r != shard:
            rc = lib.mcx_copy_to_device(vp(ptr + r * slot * 4), vp(host.ctypes.data + r * slot * 4), slot * 4, vp(st))
            if rc:
                return rc
    calls[0] += 1
    return 0

eng.set_exchange(exchange)
assert eng.exchange_self_check(), "rank %%d: a slot did not arrive as sent" %% rank
calls[0] = 0
vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
eng.run(nsamp, nburn, O.default_pinit(d, n, g0=rank * n), vl)
c = eng.counters
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), state=eng.state, mean=eng.mean, var=eng.var,
         musigall=eng.musigall, samples=eng.samples, calls=calls[0],
         counters=np.array([c["remote_steps"], c["remote_passes"], c["naccept_main"], c["exchanges"]], np.int64))
eng.close()
dist.barrier()
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("eager", [0, 1])
def test_two_processes_gloo_hip_engines_equal_oracle(tmp_path, eager):
    d, n, nburn, nsamp, pl = 16, 4096, 120, 60, 0.8
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path), d=d, n=n, nburn=nburn, nsamp=nsamp, pl=pl, eager=eager))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    vl, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    engs = [O.Engine(d, n, nshards=2, shard=s, pl=pl, threads=8) for s in range(2)]
    for e in engs:
        e.set_record(samples=True, mask=False)
    O.run_all(engs, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(2)], vl)
    for s in range(2):
        got = np.load(tmp_path / ("rank%d.npz" % s))
        # eager: at every SYNCSTEP-th step (src/mcpar.cc:127); lazy: only the gathers a remote step goes on to read
        assert int(got["calls"]) == int(got["counters"][3])
        assert int(got["calls"]) == nsamp // 10 if eager else 0 < int(got["calls"]) <= nsamp // 10
        assert engs[s].remote_steps > 0
        assert list(got["counters"][:3]) == [engs[s].remote_steps, engs[s].remote_passes, engs[s].naccept_main]
        for name in ("state", "mean", "var", "musigall", "samples"):
            assert np.array_equal(got[name].view(np.uint32), getattr(engs[s], name).view(np.uint32)), name
        engs[s].close()
