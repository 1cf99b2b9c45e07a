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
    """the HIP runtime libpt_hip.so itself is linked to, found in this process's mappings (opening it by name could bring
    in a SECOND copy: torch wheels bundle their own libamdhip64, see capi._share_torch_hip_runtime)"""
    from path_tracer_amd import capi as m
    m.load()
    paths = []
    with open("/proc/self/maps") as fh:
        for line in fh:
            f = line.split()
            if len(f) >= 6 and "libamdhip64.so" in f[5] and f[5] not in paths:
                paths.append(f[5])
    if len(paths) != 1:
        return None
    try:
        return C.CDLL(paths[0])
    except OSError:
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


def test_static_render_of_two_contexts_without_a_sync_in_between(capi, cameras, scenes):
    """every context on a GPU shares the static array but launches on a stream of its own.  Two pt_render_static
    calls back to back, no host synchronisation between them: the second call's schedule kernels and queue reset must wait
    for the first call's trace kernel (the library orders them with one event per device).  Several rounds, different image
    sizes and pass counts, so that an overlap would leave tile indices or radiance of the wrong call in the array."""
    sc, cam = scenes["mixed"], cameras["mixed"]
    shapes = [(640, 360, 10), (320, 176, 6)]
    want = []
    for (W, H, spp) in shapes:
        a = make(capi, sc, cam, W, H, 8)
        a.render(1, spp)
        a.render(spp + 1, 3)
        want.append(a.read_accum())
        a.close()
    ctxs = [make(capi, sc, cam, W, H, 8) for (W, H, _) in shapes]
    for c in ctxs:
        c.synchronize()
    for rounds in range(3):
        for c in ctxs:
            c.reset()
        for c, (_, _, spp) in zip(ctxs, shapes):          # enqueue both, no sync
            c.render_static(1, spp)
        for c, (_, _, spp) in zip(reversed(ctxs), reversed(shapes)):
            c.render_static(spp + 1, 3)
        for c, w in zip(ctxs, want):
            got = c.read_accum()
            assert bits_equal(got, w), (rounds, n_diff_pixels(got, w))
    for c in ctxs:
        c.close()


def test_static_render_under_hip_graph_capture(capi, cameras, scenes):
    """the claim in include/pt_hip.h: pt_render_static can be captured into a HIP graph.  A stream is put into capture, one
    pt_render_static call is recorded (it must neither allocate nor touch function attributes, events or the device), the
    graph is instantiated and launched twice (passes 1..6 each time onto a cleared image), and the image must equal
    pt_render's bit for bit both times; pt_last_render_ms afterwards has a defined answer (no timed render)."""
    hip = _hip()
    if hip is None:
        pytest.skip("HIP runtime library not loadable by name")
    vp = C.c_void_p
    hip.hipStreamCreate.argtypes = [C.POINTER(vp)]
    hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipStreamSynchronize.argtypes = [vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipStreamDestroy.argtypes = [vp]
    hip.hipGraphGetNodes.argtypes = [vp, vp, C.POINTER(C.c_size_t)]
    sc, cam = scenes["default"], cameras["bench"]
    W, H, SPP = 320, 176, 6
    a = make(capi, sc, cam, W, H, 4)
    a.render(1, SPP)
    want = a.read_accum()
    a.close()
    b = make(capi, sc, cam, W, H, 4)
    stream = vp()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    b.set_stream(stream)
    b.synchronize()
    graph, gexec = vp(), vp()
    assert hip.hipStreamBeginCapture(stream, 0) == 0            # hipStreamCaptureModeGlobal: any unsafe call in the thread fails
    try:
        b.render_static(1, SPP)
    finally:
        rc = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert rc == 0 and graph.value
    n_nodes = C.c_size_t()
    assert hip.hipGraphGetNodes(graph, None, C.byref(n_nodes)) == 0 and n_nodes.value >= 3      # queue reset, trace, accumulate at least
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    for _ in range(2):
        b.reset()                                               # on the same stream, before the graph
        assert hip.hipGraphLaunch(gexec, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        got = b.read_accum()
        assert bits_equal(got, want), n_diff_pixels(got, want)
    with pytest.raises(capi.PtError) as e:
        b.last_render_ms()
    assert e.value.code == capi.PT_E_STATE
    # the context still renders directly afterwards, on the same stream
    b.reset()
    b.render_static(1, SPP)
    assert bits_equal(b.read_accum(), want)
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
    b.set_stream(None)
    b.close()
    hip.hipStreamDestroy(stream)
