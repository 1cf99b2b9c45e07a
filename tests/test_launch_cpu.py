"""CPU: the self-launcher behind `python bench.py --gpus N` (path_tracer_amd/launch.py): ranks, environment, rank 0's stdout
passed through, failure propagation, time limit.  The children here are tiny Python programs; nothing touches a GPU."""
import io
import json
import os
import subprocess
import sys
import time

from conftest import ROOT
from path_tracer_amd import launch

CHILD = r"""
import json, os, sys, time
rank = int(os.environ["RANK"])
mode = sys.argv[1]
if mode == "env":
    print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PT_LAUNCHED", "HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
elif mode == "fail1":
    if rank == 1:
        sys.exit(7)
    time.sleep(60)            # the launcher must stop this rank once rank 1 has failed
elif mode == "hang":
    time.sleep(60)
elif mode == "gloo":
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"sum": float(t.item())}), flush=True)
    dist.destroy_process_group()
"""


def run(world, mode, **kw):
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc = launch.launch(world, [sys.executable, "-c", CHILD, mode], out=out, err=err, **kw)
    return rc, out.getvalue(), err.getvalue(), time.monotonic() - t0


def test_ranks_get_the_torch_distributed_environment():
    rc, out, err, _ = run(3, "env")
    assert rc == 0
    mine = [json.loads(l) for l in out.splitlines()]
    others = [json.loads(l.split("] ", 1)[1]) for l in err.splitlines() if l.startswith("[rank ")]
    assert len(mine) == 1 and len(others) == 2                  # rank 0 -> stdout, the others -> stderr with a prefix
    envs = sorted(mine + others, key=lambda e: int(e["RANK"]))
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["LOCAL_WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["PT_LAUNCHED"] == "1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and 1024 < int(envs[0]["MASTER_PORT"]) < 65536
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_a_failing_rank_ends_the_job_with_its_code():
    rc, out, err, dt = run(3, "fail1")
    assert rc == 7 and "rank 1 exited with 7" in err
    assert dt < 30, "the surviving ranks were not stopped"


def test_time_limit():
    rc, _, err, dt = run(2, "hang", timeout_s=1.0)
    assert rc == 124 and dt < 30 and "stopping all ranks" in err


def test_world_2_rendezvous_over_gloo():
    rc, out, err, _ = run(2, "gloo", timeout_s=120)
    assert rc == 0, err
    assert json.loads(out.strip().splitlines()[-1]) == {"sum": 3.0}


def test_bench_py_launches_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns two ranks before touching any GPU; here (no GPU)
    both ranks stop with bench.py's "needs a GPU" message and the parent returns non-zero instead of the old
    "launch with torch.distributed.run" refusal"""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_group.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GLOO_SOCKET_IFNAME"] = "lo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launch with" not in r.stderr and "[launch] rank" in r.stderr
    assert "needs a GPU" in r.stderr
