"""The library's OWN exchange hook (mcx_exchange.hip: rccl_exchange -- BEGIN records an event on the step stream, the side
stream waits for it and carries the in-place all-gather, WAIT makes the step stream wait for the gather's event;
MCX_OPT_ASYNC_TAIL leaves the run's last gather in flight; mcx_exchange_rccl_destroy while one is pending) with a REAL
peer on one GPU: `nshards` engines in `nshards` threads of this process, the collective supplied by tests/cpp/rccl_stub.hip
through MCX_RCCL_LIB (RCCL itself refuses two ranks on one device).  Started by tests/test_gpu_rccl_stub.py in a process
of its own, because a process loads one RCCL for good.  Replaces MPI_Allgather of src/mcpar.cc:127-140.

argv: nshards d n nburn nsamp pl eager async_tail runs persist workdir"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    nshards, d, n, nburn, nsamp = (int(v) for v in sys.argv[1:6])
    pl, eager, async_tail, runs, persist = float(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9]), int(sys.argv[10])
    work = sys.argv[11]
    import ctypes as C
    import mcpar_amd as M
    from mcpar_amd import engine as E
    import oracle_lib as O
    assert os.environ.get("MCX_RCCL_LIB"), "start me with MCX_RCCL_LIB=<rccl_stub.so>"
    M.load().mcx_set_device(0)
    assert E.rccl_available(), M.load().mcx_last_error()
    uid = E.rccl_unique_id()
    assert uid.startswith(b"rccl-stub-"), "the stub is not the RCCL that libmcx loaded"
    engs = [M.Engine(d, n, nshards=nshards, shard=s, pl=pl) for s in range(nshards)]
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    errs, checks = [], [None] * nshards

    def workfn(s):
        try:
            e = engs[s]
            e.rccl_init(uid)  # collective over the threads
            assert e.rccl_info() == (nshards, s)
            checks[s] = e.exchange_self_check()
            e.set_option(E.OPT_EAGER_EXCHANGE, eager)
            e.set_option(E.OPT_ASYNC_TAIL, async_tail)
            e.set_option(E.OPT_PERSIST, persist)
            for _ in range(runs):
                e.run(nsamp, nburn, O.default_pinit(d, n, g0=s * n), vl)
        except Exception as ex:  # noqa: BLE001
            errs.append((s, repr(ex)))

    th = [threading.Thread(target=workfn, args=(s,)) for s in range(nshards)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert all(checks), checks
    stub = C.CDLL(os.environ["MCX_RCCL_LIB"])
    stub.rccl_stub_allgather_calls.restype = C.c_uint64
    calls = int(stub.rccl_stub_allgather_calls())
    for s, e in enumerate(engs):
        c = e.counters  # (getters wait for a gather left in flight and make the final publish)
        np.savez(os.path.join(work, "shard%d.npz" % s), state=e.state, mean=e.mean, var=e.var, musigall=e.musigall,
                 samples=e.samples, stub_calls=calls,
                 counters=np.array([c["remote_steps"], c["remote_passes"], c["naccept_main"], c["exchanges"], c["exchange_waits"]], np.int64))
    # destroy while a gather may still be pending: shard 0 runs once more and is torn down right behind mcx_run; its peers
    # must take part in that run's collectives, then everybody closes
    def again(s):
        try:
            engs[s].run(nsamp, nburn, O.default_pinit(d, n, g0=s * n), vl)
            if s == 0:
                engs[s].rccl_destroy()
            engs[s].close()
        except Exception as ex:  # noqa: BLE001
            errs.append((s, repr(ex)))
    th = [threading.Thread(target=again, args=(s,)) for s in range(nshards)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    print("rccl stub worker ok: %d all-gather calls" % calls)


if __name__ == "__main__":
    main()
