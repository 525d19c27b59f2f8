"""The exchange the library ships: in-place ncclAllGather of the musigall slots (mcx_exchange_rccl_*), the
replacement of MPI_Allgather(MPI_IN_PLACE, ..., musigall) at src/mcpar.cc:127-140.

RCCL refuses a communicator with two ranks on one GPU, so on a one-GPU box only the single-rank communicator
can run (dlopen of librccl, unique id, ncclCommInitRank, the in-place all-gather on the side stream, the event
hand-shake with the engine's stream); the two-process test needs two visible GPUs and says so when it skips."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_rccl_is_loadable_and_single_rank_gather_runs():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    assert E.rccl_available(), M.load().mcx_last_error()
    uid = E.rccl_unique_id()
    assert len(uid) == 128 and uid != bytes(128)
    d, n = 16, 2048
    eng = M.Engine(d, n, pl=0.8)
    eng.rccl_init(uid)
    eng.debug_exchange()  # BEGIN (side stream, behind the engine's stream) + WAIT + drain
    assert eng.rccl_info() == (1, 0)
    assert eng.exchange_self_check()  # slot filled with shard + 1, gathered, read back
    p = O.default_pinit(d, n)
    vg, k1 = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng.run(60, 120, p, vg)
    vo, k2 = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eo = O.Engine(d, n, pl=0.8)
    eo.run(60, 120, p, vo)
    for name in ("state", "mean", "var", "musigall"):
        assert same_bits(getattr(eng, name), getattr(eo, name)), name
    eng.debug_exchange()  # the gather of a 1-rank communicator leaves the slot as it is
    assert same_bits(eng.musigall, eo.musigall)
    eng.rccl_destroy()
    eng.rccl_destroy()  # idempotent
    with pytest.raises(M.McxError):
        eng.debug_exchange()  # no hook installed any more
    eng.close()


def test_rccl_rejects_a_communicator_of_the_wrong_size():
    import mcpar_amd as M
    from mcpar_amd import engine as E
    eng = M.Engine(8, 64, nshards=2, shard=0)
    # a 2-shard engine cannot adopt a 1-rank communicator: build one through a 1-shard engine's id ... the
    # init itself would wait for the second rank, so only the argument checks are exercised here
    with pytest.raises(M.McxError):
        eng.rccl_init_raw(None)
    with pytest.raises(M.McxError):
        eng.run(10, 10, O.default_pinit(8, 64), M.make_vlfunc(M.VL_ROSENBROCK1, 8)[0])  # nshards > 1 needs an exchange
    eng.close()


@pytest.mark.parametrize("eager", [0, 1], ids=["lazy", "eager"])
def test_two_processes_two_gpus_rccl_equals_oracle(tmp_path, eager):
    import mcpar_amd as M
    from mcpar_amd import engine as E
    ndev = E.device_count()
    if ndev < 2:
        pytest.skip("needs 2 visible GPUs (this box has %d): RCCL refuses two ranks on one device" % ndev)
    d, n, nshards, nburn, nsamp, pl = 16, 4096, 2, 150, 95, 0.85
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_worker.py"), str(r), str(nshards), str(d), str(n),
                               str(nburn), str(nsamp), str(pl), str(eager), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(nshards)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode("utf-8", "replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    vo, keep = O.make_vlfunc(O.VL_ROSENBROCK1, d)
    eos = [O.Engine(d, n, nshards=nshards, shard=s, pl=pl) for s in range(nshards)]
    for e in eos:
        e.set_record(samples=True, mask=False)
    O.run_all(eos, nsamp, nburn, [O.default_pinit(d, n, g0=s * n) for s in range(nshards)], vo)
    for s in range(nshards):
        got = np.load(os.path.join(tmp_path, "shard%d.npz" % s))
        eo = eos[s]
        assert list(got["counters"][:3]) == [eo.remote_steps, eo.remote_passes, eo.naccept_main]
        assert not eager or got["counters"][3] == (nsamp + 9) // 10
        for name in ("state", "mean", "var", "samples", "musigall"):
            assert same_bits(got[name], getattr(eo, name)), (s, name)
