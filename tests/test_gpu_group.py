"""pt_group_* (include/pt_hip.h): the native RCCL sample split.  A one-GPU box can only hold a group of one rank (RCCL
refuses two ranks on one device), so this covers the whole call sequence -- id, communicator, pass split, in-place
reduce on the context stream, scaling on rank 0 -- at world = 1; the split arithmetic itself is shared with
path_tracer_amd/distributed.py, whose world-2 behaviour tests/test_distributed_cpu.py checks over gloo."""
import numpy as np
import pytest

from conftest import SCENE_CAM  # noqa: F401  (fixtures come from conftest)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as m
    m.load()
    return m


def test_group_of_one_equals_sum_then_scale(capi, cameras, scenes):
    W, H, SPP = 128, 72, 24
    sc, cam = scenes["default"], cameras["bench"]
    a = capi.Renderer(W, H, device=0)
    a.upload_scene(sc)
    a.SetUniformCamera(cam)
    a.SetUniform1i("maxBounces", 4)
    a.SetUniform1i("debugNormal", 0)
    # before pt_group_init: a state error, not a crash
    with pytest.raises(capi.PtError):
        a.group_render(1, SPP)
    uid = capi.Renderer.group_unique_id()
    assert len(uid) == 128
    a.group_init(uid, 0, 1)
    with pytest.raises(capi.PtError):
        a.group_init(uid, 0, 1)           # already in a group
    a.group_render(1, SPP)
    got = a.read_accum()
    a.group_render(1, 8)                  # a second, smaller collective on the same communicator
    got8 = a.read_accum()
    a.group_render(5, 3)                  # few passes: the short-launch kernels (running sum in registers)
    got3 = a.read_accum()
    a.group_finalize()
    a.group_finalize()                    # idempotent
    a.reset()
    a.render_sum(1, SPP)
    a.scale_accum(1.0 / SPP)
    want = a.read_accum()
    a.reset()
    a.render_sum(1, 8)
    a.scale_accum(1.0 / 8)
    want8 = a.read_accum()
    a.reset()
    a.render_sum(5, 3)
    a.scale_accum(1.0 / 3)
    want3 = a.read_accum()
    a.close()
    assert np.all(got[:, :, 3] == 1.0)
    # same kernels, same order of additions: bit-identical, not just close
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got8.view(np.uint32), want8.view(np.uint32))
    assert np.array_equal(got3.view(np.uint32), want3.view(np.uint32)) and np.all(got3[:, :, 3] == 1.0)


def test_group_bad_arguments(capi):
    a = capi.Renderer(64, 40)
    uid = capi.Renderer.group_unique_id()
    for rank, world in [(-1, 1), (1, 1), (0, 0)]:
        with pytest.raises(capi.PtError):
            a.group_init(uid, rank, world)
    a.close()


def test_group_render_reports_a_local_failure_after_the_agreement_round(capi, cameras, scenes):
    """a rank whose scene is missing (or whose arguments are bad) must not leave the collective alone: the status travels
    with the arguments in one all-reduce, then every rank returns an error (at world 1: this rank, without hanging)"""
    a = capi.Renderer(64, 40)
    a.group_init(capi.Renderer.group_unique_id(), 0, 1)
    with pytest.raises(capi.PtError) as e:
        a.group_render(1, 4)               # no scene uploaded
    assert e.value.code == capi.PT_E_STATE
    with pytest.raises(capi.PtError) as e:
        a.group_render(0, 4)               # pass 0
    assert e.value.code == capi.PT_E_INVALID
    a.upload_scene(scenes["default"])
    a.SetUniformCamera(cameras["bench"])
    a.group_render(1, 4)                   # the communicator is still usable
    assert np.all(a.read_accum()[:, :, 3] == 1.0)
    a.group_finalize()
    a.close()


def _bench(args, extra_env, timeout):
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_py_split_path_over_rccl_at_world_one():
    """the N > 1 code path of bench.py -- process group on RCCL, sum render of this rank's passes, ONE reduce, scale on rank 0 --
    with the only world size a one-GPU box can give RCCL (PT_BENCH_FORCE_SPLIT=1): not a measurement, the path end to end"""
    r, d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1"], {"PT_BENCH_FORCE_SPLIT": "1"}, 300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"] and d["config"]["spp_total"] == 64
    assert d["value"] > 0 and d["roofline"]["work_per_launch"]["samples"] == 1920 * 1080 * 64
    assert 0 < d["config"]["reduce_fraction_of_step"] < 1


def test_bench_py_self_launches_two_ranks_in_rehearsal_mode():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns the ranks before any GPU call.  On a one-GPU box
    both ranks share cuda:0 and gloo stands in for RCCL (PT_BENCH_REHEARSAL=1): the N > 1 code path end to end -- pass split,
    sum render, one reduce, scale on rank 0, max-over-ranks timing, one line from rank 0 -- not a measurement.
    Round 2 saw this stall in most runs.  Located in round 3 (tools/rehearsal_ab.sh, tools/stall_probe.py): it
    takes BOTH two processes with kernels in flight on the one device at the same time AND gloo's own staging of a device
    tensor (a high-priority pool stream waiting on the render stream, pinned copies, a worker thread synchronising that
    stream); either alone runs clean.  The rehearsal now orders both explicitly -- the ranks take the device in turn and the
    image crosses to the host with one plain copy before a host-tensor reduce -- so a stall here is a failure, not a skip."""
    r, d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"],
                  {"PT_BENCH_REHEARSAL": "1", "GLOO_SOCKET_IFNAME": "lo", "PT_LAUNCH_TIMEOUT": "60", "PT_BENCH_TRACEBACK_ON_TERM": "1"}, 150)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["config"]["spp_total"] == 128 and "REHEARSAL" in d["data"]
    assert d["value"] > 0 and d["roofline"]["work_per_launch"]["samples"] == 1920 * 1080 * 64
    assert d["ms_per_step"] < 1000, "two 6 ms renders taken in turn plus a 33 MB host reduce: tens of milliseconds"
