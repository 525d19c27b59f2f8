import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("gloo", rank=0, world_size=1)
import numpy as np
import mcpar_amd as M
from mcpar_amd import engine as E
import oracle_lib as O
print("rccl available:", E.rccl_available(), M.load().mcx_last_error())
uid = E.rccl_unique_id()
eng = M.Engine(16, 2048, pl=0.8)
eng.rccl_init(uid)
eng.debug_exchange()
vg,_ = M.make_vlfunc(M.VL_ROSENBROCK1, 16)
eng.run(60, 120, O.default_pinit(16, 2048), vg)
print("ok", eng.counters["remote_passes"])
import ctypes
print([l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l or "libamdhip64" in l][:6])
