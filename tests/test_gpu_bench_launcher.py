"""`python bench.py --gpus 2` with no launcher around it (the driver's round-end command line, BENCH_r04.cmd) runs TWO ranks
-- here both on the box's one GPU (--one-device: the host-staged exchange, RCCL refuses two ranks on one device) -- and the
line says n_gpus 2, which exchange ran and why.  src/mcpar.cc:127-140,225."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_started_directly_is_a_two_rank_run():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--steps", "1", "--warmup", "1",
                        "--no-extras"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=500)
    assert r.returncode == 0, r.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["scaling"] == "weak"
    assert o["launcher"]["ranks"] == 2
    c = o["config"]
    assert len(c["pci_bus_ids"]) == 2
    assert c["rccl_comm_ranks"] is None and "requested" in c["exchange_backend"]
    assert o["value"] > 0 and list(o)[-1] == "summary"


def test_gpus_2_without_one_device_on_a_one_gpu_box_refuses():
    import mcpar_amd  # noqa: F401
    from mcpar_amd import engine as E
    if E.device_count() >= 2:
        pytest.skip("this box has two GPUs: the refusal is for boxes with fewer GPUs than ranks")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 66
    assert r.stdout.decode().strip() == "" and "refusing to run" in r.stderr.decode()
