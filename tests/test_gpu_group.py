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
