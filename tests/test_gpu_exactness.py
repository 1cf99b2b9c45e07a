"""What a bounding box cannot stand for: the float triangle test of rt.comp:153-173 divides by a determinant
that is rounding noise for a triangle whose edges at p1 are parallel, and then reports hits anywhere along the triangle's
line.  The traversal kernels must return them too -- bit for bit what the brute-force kernel (every triangle behind the
root-AABB gate, like the shader) returns.  The first case is the render tools/fuzz_bvh.py found (seed 10825: a zero-area
triangle BEHIND the camera wins a pixel with t = 1.0); the others are seeded samples of the same generators."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT, bits_equal, n_diff_pixels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as m
    m.load()
    return m


@pytest.fixture(scope="module")
def fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_bvh", os.path.join(ROOT, "tools", "fuzz_bvh.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def make(capi, sc, cam, W, H, bounces, variant, lds_block=0):
    r = capi.Renderer(W, H)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    r.set_tuning(0, 0, 0, variant)
    if lds_block:
        r.set_option("lds_block", lds_block)
    r.set_option("no_fused", 1)
    return r


def test_zero_area_triangle_behind_the_camera_wins_its_pixel(capi):
    """the render tools/fuzz_bvh.py caught (1,500-triangle soup, seed 10825 of the generator as it was then): triangle
    p2 == p3 lies BEHIND the camera, and the shader's arithmetic gives it pixel (23, 10) of pass 2 at t = 1.0"""
    from path_tracer_amd import scene as S
    f = np.float32
    bad = [[f(-0.08934768), f(0.9112981), f(-0.10837321)], [f(0.11877692), f(0.6685527), f(-0.3100864)], [f(0.11877692), f(0.6685527), f(-0.3100864)]]
    corners = [[[-1.5, -1.2, -1.5], [-1.4, -1.2, -1.5], [-1.5, -1.1, -1.5]], [[1.9, 1.9, 1.7], [1.8, 1.9, 1.7], [1.9, 1.8, 1.7]]]   # the soup's bounds
    bunny = S.bunny_mesh(3).tris
    tris = np.concatenate([bunny[:100], S.triangles_from_points(np.array([bad], f), 1), bunny[100:], S.triangles_from_points(np.array(corners, f), 3)])
    rc, in_tree, listed, never, _ = capi.classify_mesh(tris)
    assert rc == 0 and (listed, never) == (1, 0) and in_tree == len(tris) - 1
    sc = S.default_scene(S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 3))
    cam = S.Camera(position=np.array([0.21326542, 0.33497703, 0.07151908], f), yaw=312.03761102568546, pitch=-8.1856729091284).UpdateView()
    W, H, SPP, B = 200, 113, 5, 4
    ref = make(capi, sc, cam, W, H, B, 1)
    ref.render(1, SPP)
    want = ref.read_accum()
    ids, ts, _, _ = ref.read_primary_hits(2)
    assert ids[10, 23] == 100 and ts[10, 23] == 1.0, "the spurious hit the shader's arithmetic reports (pass 2, pixel 23,10)"
    ref.close()
    for variant, blk in ((5, 0), (6, 512), (6, 1024)):
        r = make(capi, sc, cam, W, H, B, variant, blk)
        r.render(1, SPP)
        assert r.kernel_name().startswith("pt_trace_items_bvh"), "the traversal must be what is tested"
        got = r.read_accum()
        assert bits_equal(got, want), (variant, blk, n_diff_pixels(got, want))
        ids, ts, _, _ = r.read_primary_hits(2)
        assert ids[10, 23] == 100 and ts[10, 23] == 1.0
        r.close()


@pytest.mark.parametrize("seed", [3, 17, 29])
def test_bunny_with_slivers_and_zero_area_triangles(capi, fuzz, cameras, seed):
    """bunny1 + 30 slivers (sine of the angle at p1 from 1e-7 to 0.1) + 12 zero-area triangles + 6 with a zero edge at p1:
    some boxes grown, some triangles in the always-reached leaves, some dropped -- and the image equal to brute force's"""
    from path_tracer_amd import scene as S
    rng = np.random.default_rng(seed)
    p = fuzz.soup(rng, 48, (0.0, 0.4, 0.0), 0.8, 0.05, 0.6).astype(np.float64)
    e1 = p[:30, 1] - p[:30, 0]
    perp = np.cross(e1, rng.normal(0, 1, (30, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    ang = np.exp(rng.uniform(np.log(1e-7), np.log(0.1), (30, 1)))
    p[:30, 2] = p[:30, 0] + (e1 + perp * np.linalg.norm(e1, axis=1, keepdims=True) * ang) * rng.uniform(0.3, 1.5, (30, 1))
    p[30:42, 2] = p[30:42, 1]
    p[42:48, 1] = p[42:48, 0]
    extra = S.triangles_from_points(p.astype(np.float32), 1)
    bunny = S.bunny_mesh(3).tris
    tris = np.concatenate([bunny[:300], extra, bunny[300:]])           # in the middle: canonical ids on both sides of them
    rc, in_tree, listed, never, grow = capi.classify_mesh(tris)
    assert rc == 0 and never == 6 and 12 <= listed <= 42 and in_tree + listed + never == len(tris)
    sc = S.default_scene(S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 3))
    cam = cameras["bench"]
    W, H, SPP, B = 320, 176, 6, 6
    ref = make(capi, sc, cam, W, H, B, 1)
    ref.render(1, SPP)
    want = ref.read_accum()
    ref.close()
    for variant, blk in ((5, 0), (6, 512), (6, 1024)):
        r = make(capi, sc, cam, W, H, B, variant, blk)
        r.render(1, SPP)
        assert r.kernel_name().startswith("pt_trace_items_bvh")
        got = r.read_accum()
        assert bits_equal(got, want), (variant, blk, n_diff_pixels(got, want))
        r.close()


def test_a_mesh_of_slivers_goes_to_the_brute_force_kernels(capi, fuzz, cameras):
    """more triangles that every ray has to test than 64 and than half of the mesh: the tree is refused, the image is the
    brute-force one; with fewer of them than half the mesh the tree is used and must give the same image"""
    from path_tracer_amd import scene as S
    rng = np.random.default_rng(5)
    p = fuzz.soup(rng, 400, (0.0, 0.4, 0.0), 0.8, 0.05, 0.4)
    zero = p.copy()
    zero[:, 2] = zero[:, 1]                                # p2 == p3: parallel edges
    cam, W, H, B = cameras["bench"], 160, 88, 4
    for n_zero, tree in ((300, False), (150, True)):
        tris = S.triangles_from_points(np.concatenate([zero[:n_zero], p[n_zero:]]), 1)
        rc, in_tree, listed, never, _ = capi.classify_mesh(tris)
        assert (in_tree, listed, never) == (400 - n_zero, n_zero, 0)
        sc = S.default_scene(S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 1))
        ref = make(capi, sc, cam, W, H, B, 1)
        ref.render(1, 3)
        want = ref.read_accum()
        ref.close()
        r = make(capi, sc, cam, W, H, B, 0)
        r.render(1, 3)
        assert ("bvh" in r.kernel_name()) == tree, r.kernel_name()
        assert bits_equal(r.read_accum(), want)
        r.close()


@pytest.mark.parametrize("first,kind,far", [(0, None, False), (20000, 9, False), (40000, 6, False), (30000, 10, False), (50000, None, True)])
def test_seeded_random_scenes_traversal_equals_brute_force(capi, fuzz, first, kind, far):
    """a slice of tools/fuzz_bvh.py (the round's campaign: 10,400 scene renders, profiles/r3_exactness): 12 scenes per generator"""
    for seed in range(first, first + 12):
        sc, cam, bounces = fuzz.random_scene(seed, kind, True if far else None)
        want, _ = fuzz.render(sc, cam, 128, 72, bounces, 3, 1, 0, True)
        for variant, blk in ((5, 0), (6, 0)):
            got, name = fuzz.render(sc, cam, 128, 72, bounces, 3, variant, blk, True)
            assert bits_equal(got, want), (seed, variant, name, n_diff_pixels(got, want))


@pytest.mark.parametrize("preset", [0, 1])
def test_seeded_random_scenes_against_the_oracle(capi, fuzz, preset):
    """a slice of tools/fuzz_oracle.py (the round's campaign: 10,970 scenes, 0 differences): the default kernel choice and the
    brute-force kernel against oracle/pt_oracle.c on scenes far from the fixtures"""
    import oracle_py as O
    n = 0
    for seed in range(500000 + 1000 * preset, 500000 + 1000 * preset + 40):
        sc, cam, bounces = fuzz.random_scene(seed)
        if len(sc.mesh_tris) > 2000:
            continue
        want = O.Oracle(sc, cam, 96, 56, bounces, preset=preset).render(1, 2)
        for variant in (0, 1):
            got, name = fuzz.render(sc, cam, 96, 56, bounces, 2, variant, 0, False, preset)
            assert bits_equal(got, want), (seed, variant, name, n_diff_pixels(got, want))
        n += 1
    assert n >= 25


def test_seeded_scheduling_cases_equal_the_frame_loop(capi):
    """a slice of tools/fuzz_sched.py (the round's campaign: 4,400 cases x 3 routes, 0 differences): random image sizes, pass
    ranges, passes per work item, workspace limits, waves per CU, scheduling options and kernel variants through pt_render,
    pt_render_static and a split pair of pt_render calls, against one brute-force dispatch per pass"""
    spec = importlib.util.spec_from_file_location("fuzz_sched", os.path.join(ROOT, "tools", "fuzz_sched.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sc_list = m.scenes()
    for case in range(40):
        assert m.run_case(case, sc_list) == 0, case


def test_point_sphere_at_the_origin_seen_from_the_origin(capi):
    """the sphere margin's k D^2 / (sqrt(r^2 + k D^2) + r) with r = 0 and D = 0 must not turn the traversal's margins into NaN:
    radius-0 spheres at the origin, the camera at the origin (the shader reports a hit at t = 0 for every ray: the tree must too)"""
    from path_tracer_amd import scene as S
    sc = S.Scene(materials=S.default_materials())
    sc.add_sphere((0.0, 0.0, 0.0), 0.0, 1)
    for k in range(40):
        sc.add_sphere((0.0, 0.0, 0.0), 0.0, 1)             # enough records for a sphere tree
    sc.spheres["position"][1:] = 0.0
    sc.spheres["position"][1, 0] = 2.0                     # one real sphere in front of the camera (yaw 0 looks down +x)
    sc.spheres["radius"][1] = 0.5
    cam = S.Camera(position=np.zeros(3, np.float32), yaw=0.0, pitch=0.0).UpdateView()
    W, H = 96, 56
    ref = make(capi, sc, cam, W, H, 3, 1)
    ref.render(1, 2)
    want = ref.read_accum()
    ids, ts, _, _ = ref.read_primary_hits(1)
    # rt.comp:126-146 accepts `t1 > -HIT_LIMIT`: the point sphere the camera sits in is hit by every primary ray at t = 0
    assert (ids == 0).all() and (ts == 0.0).all()
    ref.close()
    r = make(capi, sc, cam, W, H, 3, 0)
    r.render(1, 2)
    assert "bvh" in r.kernel_name()
    assert bits_equal(r.read_accum(), want)
    r.close()


@pytest.mark.parametrize("scale", [1e-12, 1e-6, 1e6, 1e16])
def test_seeded_random_scenes_at_extreme_scales(capi, fuzz, scale):
    """every coordinate, radius and the camera position times a power of ten: the slab products near overflow (1e16) or the
    triangle test's determinants in the denormals (1e-12); the quantised node's folded decoding (2^e * inv, o * inv + a) included"""
    for seed in range(2000000, 2000012):
        sc, cam, bounces = fuzz.random_scene(seed, None, None, False, scale)
        want, _ = fuzz.render(sc, cam, 128, 72, bounces, 3, 1, 0, True)
        for variant in (5, 6):
            got, name = fuzz.render(sc, cam, 128, 72, bounces, 3, variant, 0, True)
            assert bits_equal(got, want), (seed, variant, name, n_diff_pixels(got, want))
