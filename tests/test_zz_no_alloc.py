"""pt_render_static (include/pt_hip.h): pt_render for callers that cannot allocate -- queue, tile schedule and per-pass
workspace live in a static device array of the library.  Through the C ABI, against pt_render, bit for bit: two calls in a
row (the second on a smaller image with fewer passes, so stale contents of the shared array would show), the short-launch
route, the route taken when the array holds too few passes of the image, the size limit, and that device memory in use does
not move across the calls."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits_equal, n_diff_pixels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as m
    m.load()
    return m


def _hip():
    """the HIP runtime already in the process (for hipMemGetInfo only)"""
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return C.CDLL(name)
        except OSError:
            pass
    return None


def _used_bytes():
    h = _hip()
    if h is None:
        return None
    free, total = C.c_size_t(), C.c_size_t()
    if h.hipMemGetInfo(C.byref(free), C.byref(total)) != 0:
        return None
    return total.value - free.value


def make(capi, sc, cam, W, H, bounces):
    r = capi.Renderer(W, H)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    return r


def test_static_render_equals_pt_render_twice_in_a_row(capi, cameras, scenes):
    sc, cam = scenes["mixed"], cameras["mixed"]
    want, got = {}, {}
    for (W, H, spp) in ((320, 176, 12), (160, 88, 7)):
        a = make(capi, sc, cam, W, H, 6)
        a.render(1, spp)
        want[(W, H)] = a.read_accum()
        a.close()
    b = make(capi, sc, cam, 320, 176, 6)
    b.synchronize()
    used0 = _used_bytes()
    b.render_static(1, 12)                       # workspace route: 12 passes x 320 x 176 x 12 B in the static array
    got[(320, 176)] = b.read_accum()
    c = make(capi, sc, cam, 160, 88, 6)          # a second context, a smaller image, fewer passes: shares the array
    used1 = _used_bytes()
    c.render_static(1, 7)
    got[(160, 88)] = c.read_accum()
    c.synchronize()
    used2 = _used_bytes()
    for k in want:
        assert bits_equal(got[k], want[k]), (k, n_diff_pixels(got[k], want[k]))
    # continuing an accumulation, and the first context again after the other one has used the array
    b.render_static(13, 3)
    a = make(capi, sc, cam, 320, 176, 6)
    a.render(1, 12)
    a.render(13, 3)
    assert bits_equal(b.read_accum(), a.read_accum())
    if used0 is not None:
        assert used2 == used1, "device memory in use moved across pt_render_static"
    a.close(); b.close(); c.close()


def test_static_render_short_launch_and_default_scene(capi, cameras, scenes):
    sc, cam = scenes["default"], cameras["bench"]
    a = make(capi, sc, cam, 256, 144, 4)
    b = make(capi, sc, cam, 256, 144, 4)
    for p in range(1, 4):
        a.render_pass(p)
        b.render_static(p, 1)                    # one pass per call, like the reference's frame loop
    assert bits_equal(a.read_accum(), b.read_accum())
    a.reset(); b.reset()
    a.render(1, 2); b.render_static(1, 2)
    assert bits_equal(a.read_accum(), b.read_accum())
    # brute-force variant (no fused route, workspace route only)
    a.reset(); b.reset()
    a.set_tuning(0, 0, 0, 4); b.set_tuning(0, 0, 0, 4)
    a.render(1, 9); b.render_static(1, 9)
    assert bits_equal(a.read_accum(), b.read_accum())
    a.close(); b.close()


def test_static_render_when_the_array_holds_few_passes(capi, cameras, scenes):
    """1920x1080: one pass is 24.9 MB, the array holds two -> the passes are taken in short fused launches"""
    sc, cam = scenes["default"], cameras["bench"]
    a = make(capi, sc, cam, 1920, 1080, 4)
    b = make(capi, sc, cam, 1920, 1080, 4)
    a.render(1, 10)
    b.render_static(1, 10)
    assert bits_equal(a.read_accum(), b.read_accum())
    a.close(); b.close()


def test_static_render_limits_and_state(capi, cameras, scenes):
    r = capi.Renderer(64, 40)
    with pytest.raises(capi.PtError) as e:
        r.render_static(1, 1)                    # nothing uploaded
    assert e.value.code == capi.PT_E_STATE
    r.upload_scene(scenes["default"])
    r.SetUniformCamera(cameras["bench"])
    with pytest.raises(capi.PtError) as e:
        r.render_static(0, 1)
    assert e.value.code == capi.PT_E_INVALID
    r.render_static(1, 1)
    r.close()
    big = make(capi, scenes["default"], cameras["bench"], 4096, 2304, 2)      # 64 x 2304 = 147,456 tiles
    with pytest.raises(capi.PtError) as e:
        big.render_static(1, 1)
    assert e.value.code == capi.PT_E_LIMIT and "tiles" in str(e.value)
    big.render(1, 1)                             # the allocating entry point still takes it
    big.close()
