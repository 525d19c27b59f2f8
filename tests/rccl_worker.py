"""One shard of a multi-process run over the library's own RCCL exchange (mcx_exchange_rccl_*): process `rank`
drives GPU `rank`.  Started by tests/test_gpu_rccl.py; rank 0 draws the ncclUniqueId and ships it through a file."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, nshards, d, n, nburn, nsamp = (int(v) for v in sys.argv[1:7])
    pl, eager, work = float(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
    import mcpar_amd as M
    from mcpar_amd import engine as E
    import oracle_lib as O
    M.load().mcx_set_device(rank)
    idfile = os.path.join(work, "rccl_id")
    if rank == 0:
        uid = E.rccl_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no RCCL id after 120 s" % rank)
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    eng = M.Engine(d, n, nshards=nshards, shard=rank, pl=pl)
    eng.rccl_init(uid)  # collective
    eng.debug_exchange()
    assert eng.rccl_info() == (nshards, rank)
    assert eng.exchange_self_check(), "rank %d: a slot did not arrive as sent" % rank
    eng.set_option(E.OPT_EAGER_EXCHANGE, eager)
    vl, keep = M.make_vlfunc(M.VL_ROSENBROCK1, d)
    eng.run(nsamp, nburn, O.default_pinit(d, n, g0=rank * n), vl)
    c = eng.counters
    np.savez(os.path.join(work, "shard%d.npz" % rank), state=eng.state, mean=eng.mean, var=eng.var,
             musigall=eng.musigall, samples=eng.samples,
             counters=np.array([c["remote_steps"], c["remote_passes"], c["naccept_main"], c["exchanges"]], np.int64))
    eng.close()


if __name__ == "__main__":
    main()
