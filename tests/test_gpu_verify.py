"""pt_set_option(ctx, "verify", n) (include/pt_hip.h): every n-th render of the traversal kernels is repeated by the brute-force
kernel from the same starting image and compared bit for bit on the device -- the library's own check of the BVH's skipping
margins on whatever scene the caller has.  No counterpart in the reference (which has no BVH that runs)."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as m
    m.load()
    return m


def make(capi, sc, cam, W, H, bounces):
    r = capi.Renderer(W, H)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    return r


@pytest.mark.parametrize("tag,camtag,bounces", [("default", "bench", 8), ("mixed", "mixed", 16)])
def test_verified_renders_pass_and_leave_the_same_image(capi, cameras, scenes, tag, camtag, bounces):
    sc, cam = scenes[tag], cameras[camtag]
    a = make(capi, sc, cam, 320, 176, bounces)
    b = make(capi, sc, cam, 320, 176, bounces)
    b.set_option("verify", 2)                     # every second call
    for (first, n) in ((1, 6), (7, 1), (8, 5), (13, 2)):      # long launches, one pass (fused route), continuing accumulation
        a.render(first, n)
        b.render(first, n)
    assert bits_equal(a.read_accum(), b.read_accum())
    assert b.kernel_name().startswith("pt_trace_items_bvh")      # the last launch of a checked call is the traversal kernel's
    assert b.last_render_ms() > 0
    # the sum entry point is checked too
    a.reset(); b.reset()
    b.set_option("verify", 1)
    a.render_sum(1, 5); b.render_sum(1, 5)
    assert bits_equal(a.read_accum(), b.read_accum())
    ca, cb = a.counters(), b.counters()
    assert cb["samples"] > ca["samples"]                          # counters count both renders of a checked call
    a.close(); b.close()


def test_a_wrong_tree_is_caught(capi, cameras, scenes):
    """forced mismatch: the test hook `verify_inject` uploads the tree with the root's child boxes shrunk to half their size, so
    the traversal skips triangles it must not skip.  Without `verify` the render returns PT_OK with a wrong image; with it the
    call returns PT_E_STATE, names the count, and leaves the brute-force (exact) image in place."""
    sc, cam = scenes["default"], cameras["bench"]
    good = make(capi, sc, cam, 320, 176, 6)
    good.render(1, 4)
    want = good.read_accum()
    good.close()
    r = make(capi, sc, cam, 320, 176, 6)
    r.set_option("verify_inject", 1)
    r.render(1, 4)                                # unchecked: nothing notices
    assert not bits_equal(r.read_accum(), want)
    r.reset()
    r.set_option("verify", 1)
    with pytest.raises(capi.PtError) as e:
        r.render(1, 4)
    assert e.value.code == capi.PT_E_STATE and "pixels differ" in str(e.value)
    assert bits_equal(r.read_accum(), want)       # the exact image
    # the brute-force kernels themselves are never "checked" (nothing to compare with): variant 4 renders quietly
    r.reset()
    r.set_tuning(0, 0, 0, 4)
    r.render(1, 4)
    assert bits_equal(r.read_accum(), want)
    # the fault removed: the tree is uploaded again and the check passes
    r.set_tuning(0, 0, 0, 6)
    r.set_option("verify_inject", 0)
    r.reset()
    r.render(1, 4)
    assert bits_equal(r.read_accum(), want)
    r.close()


def test_option_values_are_validated(capi):
    r = capi.Renderer(64, 40)
    for name, bad in (("lds_block", 256), ("lds_block", 768), ("lds_block", 2048), ("verify", -1), ("verify_inject", 2), ("park_exit", 9), ("nonsense", 0)):
        with pytest.raises(capi.PtError) as e:
            r.set_option(name, bad)
        assert e.value.code == capi.PT_E_INVALID, (name, bad)
    for name, ok in (("lds_block", 0), ("lds_block", 512), ("lds_block", 1024), ("verify", 0), ("verify", 1000)):
        r.set_option(name, ok)
    r.close()
