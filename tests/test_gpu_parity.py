"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) the committed fixtures from the
reference shader under llvmpipe and (b) the oracle on the same inputs.  Bar: bit-exact everywhere
(the path is float32 with a pinned operation order); the only toleranced test is the multi-GPU sum
mode, whose summation order differs from the reference's running mean by construction."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLD, SCENE_CAM, bits_equal, crc, n_diff_pixels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as m
    m.load()
    return m


def make(capi, sc, cam, W, H, bounces, debug_normal=False):
    r = capi.Renderer(W, H, device=0)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    r.SetUniform1i("debugNormal", int(debug_normal))
    return r


@pytest.mark.parametrize("tag", ["default", "default_rot", "mixed"])
@pytest.mark.parametrize("size", [(64, 40), (640, 360), (1920, 1080)])
@pytest.mark.parametrize("p", [1, 7])
def test_primary_hits_vs_reference_fixtures(capi, tag, size, p, meta, cameras, scenes):
    W, H = size
    r = make(capi, scenes[SCENE_CAM[tag][0]], cameras[SCENE_CAM[tag][1]], W, H, 3)
    ids, t, dirs, st = r.read_primary_hits(p)
    e = meta[f"primary_{tag}_{W}x{H}_p{p}"]
    assert int((ids >= 0).sum()) == e["n_hits"]
    assert crc(ids) == e["crc_ids"], "primary-hit primitive ids must be bit-exact"
    assert crc(st) == e["crc_rng_after_jitter"], "hash RNG stream must be reproduced exactly"
    assert crc(dirs) == e["crc_dirs"] and crc(t) == e["crc_t"]
    r.close()


def test_primary_hits_4k(capi, meta, cameras, scenes):
    r = make(capi, scenes["default"], cameras["bench"], 3840, 2160, 3)
    ids, t, dirs, st = r.read_primary_hits(1)
    e = meta["primary_default_3840x2160_p1"]
    assert crc(ids) == e["crc_ids"] and crc(st) == e["crc_rng_after_jitter"] and crc(dirs) == e["crc_dirs"] and crc(t) == e["crc_t"]
    r.close()


@pytest.mark.parametrize("tag", ["default", "mixed"])
def test_debug_normal(capi, tag, meta, cameras, scenes):
    g = np.load(os.path.join(GOLD, "debug_normal.npz"))
    r = make(capi, scenes[SCENE_CAM[tag][0]], cameras[SCENE_CAM[tag][1]], 128, 72, 3, debug_normal=True)
    r.render_pass(1)
    assert n_diff_pixels(r.read_accum()[:, :, :3], g[f"{tag}_128x72"]) == 0
    r.close()
    r = make(capi, scenes[SCENE_CAM[tag][0]], cameras[SCENE_CAM[tag][1]], 640, 360, 3, debug_normal=True)
    r.render_pass(1)
    assert crc(r.read_accum()[:, :, :3]) == meta[f"normal_{tag}_640x360"]["crc_rgb"]
    r.close()


@pytest.mark.parametrize("tag,bounces", [("default", 3), ("default", 4), ("default", 8), ("default_rot", 4), ("mixed", 3), ("mixed", 16)])
def test_running_mean_vs_reference_fixtures(capi, tag, bounces, cameras, scenes):
    """one dispatch per pass, like the reference's frame loop (src/Main.cpp:144-150)"""
    g = np.load(os.path.join(GOLD, "radiance_128x72.npz"))
    r = make(capi, scenes[SCENE_CAM[tag][0]], cameras[SCENE_CAM[tag][1]], 128, 72, bounces)
    for k in range(1, 17):
        r.render_pass(k)
        if k in (1, 2, 3, 4, 16):
            acc = r.read_accum()
            assert n_diff_pixels(acc[:, :, :3], g[f"{tag}_b{bounces}_128x72_mean{k}"]) == 0, f"running mean after pass {k}"
            assert np.all(acc[:, :, 3] == 1.0)
    r.close()


def test_config1_640x360_16spp_4bounces(capi, meta, cameras, scenes):
    """BASELINE.json configs[0]: the reference's own CPU-runnable case, CRC of the reference image."""
    r = make(capi, scenes["default"], cameras["bench"], 640, 360, 4)
    r.render(1, 16)
    acc = r.read_accum()
    rows = np.load(os.path.join(GOLD, "radiance_c1_rows.npz"))
    assert n_diff_pixels(acc[::8, :, :3], rows["mean16_rows8"]) == 0
    assert crc(acc[:, :, :3]) == meta["radiance_default_b4_640x360"]["crc_mean16"]
    r.close()


def test_mixed_scene_16_bounces_640x360(capi, meta, cameras, scenes):
    """BASELINE.json configs[4] scene (glass, specular, emissive; 145 spheres), 16 bounces."""
    r = make(capi, scenes["mixed"], cameras["mixed"], 640, 360, 16)
    r.render(1, 4)
    assert crc(r.read_accum()[:, :, :3]) == meta["radiance_mixed_b16_640x360"]["crc_mean4"]
    r.close()


def test_batched_render_equals_per_pass_dispatches(capi, cameras, scenes):
    a = make(capi, scenes["mixed"], cameras["mixed"], 128, 72, 8)
    b = make(capi, scenes["mixed"], cameras["mixed"], 128, 72, 8)
    a.render(1, 6)
    for k in range(1, 7):
        b.render_pass(k)
    assert bits_equal(a.read_accum(), b.read_accum())
    # resuming: passes 1..3 then 4..6
    b.reset()
    b.render(1, 3)
    b.render(4, 3)
    assert bits_equal(a.read_accum(), b.read_accum())
    a.close(); b.close()


def test_config2_1080p_64spp_8bounces_properties_and_oracle_windows(capi, cameras, scenes):
    """BASELINE.json configs[1] at full size: determinism, pass-split invariance, and bit-exact windows
    against the oracle (a full CPU render would take minutes)."""
    import oracle_py as O
    W, H, B, SPP = 1920, 1080, 8, 64
    sc, cam = scenes["default"], cameras["bench"]
    r = make(capi, sc, cam, W, H, B)
    r.render(1, SPP)
    full = r.read_accum()
    c = r.counters()
    assert c["samples"] == W * H * SPP and c["segments"] >= c["samples"] and 0 < c["mesh_segments"] < c["segments"]
    r.reset()
    r.render(1, 40)
    r.render(41, 24)
    assert crc(r.read_accum()) == crc(full), "pass-split invariance / determinism"
    assert np.all(full[:, :, 3] == 1.0) and np.isfinite(full).all()
    o = O.Oracle(sc, cam, W, H, B)
    for (x0, y0, w, h) in ((900, 380, 96, 8), (0, 0, 64, 4), (1856, 1076, 64, 4), (1000, 300, 48, 8)):
        want = o.render_region(1, SPP, x0, y0, w, h)
        assert n_diff_pixels(full[y0:y0 + h, x0:x0 + w], want) == 0, (x0, y0)
    r.close()


def test_sum_mode_matches_mean_within_tolerance(capi, cameras, scenes):
    """multi-GPU split building block: sum of per-pass radiance / spp vs the reference running mean."""
    W, H, SPP = 128, 72, 32
    a = make(capi, scenes["default"], cameras["bench"], W, H, 4)
    a.render(1, SPP)
    mean = a.read_accum()
    a.reset()
    a.render_sum(1, 20)
    a.render_sum(21, 12)
    s = a.read_accum()
    assert np.all(s[:, :, 3] == SPP)
    a.scale_accum(1.0 / SPP)
    got = a.read_accum()
    assert np.all(got[:, :, 3] == 1.0)
    # tolerance: fp32 sum of 32 terms vs running mean; radiance values are O(1)
    np.testing.assert_allclose(got[:, :, :3], mean[:, :, :3], rtol=1e-5, atol=1e-6)
    a.close()


def test_trailing_pixels_are_not_traced(capi, cameras, scenes):
    """glDispatchCompute(W/8, H/8, 1) (src/Main.cpp:148): pixels beyond the last full 8x8 group stay untouched."""
    import oracle_py as O
    W, H = 70, 45
    r = make(capi, scenes["default"], cameras["bench"], W, H, 3)
    r.render(1, 2)
    acc = r.read_accum()
    assert np.all(acc[40:, :, :] == 0) and np.all(acc[:, 64:, :] == 0)
    want = O.Oracle(scenes["default"], cameras["bench"], W, H, 3).render(1, 2)
    assert n_diff_pixels(acc[:40, :64], want[:40, :64]) == 0
    r.close()


@pytest.mark.parametrize("variant", [0, 1, 5])
def test_multi_chunk_mesh_and_tie_breaking(capi, cameras, variant):
    """mesh larger than one register-resident chunk (704 tris) + duplicated triangles (equal t: lowest index wins);
    variant 0 = library default (BVH traversal for a mesh of this size), 1 = brute force, 5 = BVH"""
    import oracle_py as O
    from path_tracer_amd import scene as S
    m = S.heightfield_mesh(24, 1)                      # 1152 triangles -> 2 chunks
    dup = np.concatenate([m.tris, m.tris[:300]])      # exact duplicates later in the list must lose ties
    mesh = S.Mesh(dup, S.Mesh.gen_bounding_box(dup), 1)
    sc = S.default_scene(mesh)
    cam = S.Camera(position=np.array([0.2, 1.6, 2.2], np.float32), yaw=95.0, pitch=35.0).UpdateView()
    W, H = 128, 72
    r = make(capi, sc, cam, W, H, 5)
    if variant:
        r.set_tuning(0, 0, 0, variant)
    ids, t, dirs, st = r.read_primary_hits(1)
    o = O.Oracle(sc, cam, W, H, 5)
    oi, ot, od, os_ = o.primary_hits(1)
    assert (oi >= 0).sum() > 2000 and np.array_equal(ids, oi) and bits_equal(t, ot)
    assert not np.any((ids >= 1152) & (ids < 1452)), "a duplicate won a tie against the earlier triangle"
    r.render(1, 3)
    assert n_diff_pixels(r.read_accum(), o.render(1, 3)) == 0
    r.close()


def test_no_mesh_spheres_only(capi, cameras):
    import oracle_py as O
    from path_tracer_amd import scene as S
    sc = S.Scene(materials=S.default_materials())
    sc.add_sphere((0.0, 0.5, 0.0), 0.5, 2)
    sc.add_sphere((0.0, -100.0, 0.0), 100.0, 1)
    sc.add_sphere((0.6, 0.3, 0.4), 0.3, 4)
    cam = cameras["bench"]
    r = make(capi, sc, cam, 64, 40, 8)
    r.render(1, 4)
    want = O.Oracle(sc, cam, 64, 40, 8).render(1, 4)
    assert n_diff_pixels(r.read_accum(), want) == 0
    r.close()


def test_error_behaviour(capi, cameras, scenes):
    L = capi.load()
    h = C.c_void_p()
    assert L.pt_create(C.byref(h), 0) == 0
    assert L.pt_render_pass(h, 1) == capi.PT_E_STATE                      # nothing set up yet
    assert b"pt_resize" in L.pt_last_error(h)
    assert L.pt_resize(h, 4, 4) == capi.PT_E_INVALID
    assert L.pt_resize(h, 64, 40) == 0
    assert L.pt_render_pass(h, 1) == capi.PT_E_STATE
    sc = scenes["default"]
    mt, nd = np.ascontiguousarray(sc.mesh_tris), np.ascontiguousarray(sc.nodes)
    assert L.pt_upload_mesh(h, mt.ctypes.data, len(mt), None, 0) == capi.PT_E_INVALID     # a mesh needs nodes[0]
    assert L.pt_upload_mesh(h, mt.ctypes.data, len(mt), nd.ctypes.data, 1) == 0
    ma = np.ascontiguousarray(sc.materials[:2])        # too few materials for the indices in use
    tr = np.ascontiguousarray(sc.triangles)
    assert L.pt_upload_scene(h, ma.ctypes.data, 2, None, 0, tr.ctypes.data, len(tr)) == 0
    cam = cameras["bench"]
    pos, iv = np.ascontiguousarray(cam.position, np.float32), cam.inv_view_flat()
    fp = C.POINTER(C.c_float)
    assert L.pt_set_camera(h, pos.ctypes.data_as(fp), iv.ctypes.data_as(fp)) == 0
    assert L.pt_render_pass(h, 1) == capi.PT_E_INVALID and b"material" in L.pt_last_error(h)
    assert L.pt_render_pass(h, 0) == capi.PT_E_INVALID                     # currAccumPass starts at 1
    assert L.pt_set_max_bounces(h, 0) == capi.PT_E_INVALID
    assert L.pt_create(C.byref(C.c_void_p()), 9999) == capi.PT_E_INVALID
    L.pt_destroy(h)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6])
def test_kernel_variants_are_bit_identical(capi, variant, meta, cameras, scenes):
    """VGPR-resident and LDS-resident brute force and the BVH traversal must give the reference image (config 1 CRC) alike"""
    r = make(capi, scenes["default"], cameras["bench"], 640, 360, 4)
    r.set_tuning(0, 0, 0, variant)
    r.render(1, 16)
    assert crc(r.read_accum()[:, :, :3]) == meta["radiance_default_b4_640x360"]["crc_mean16"]
    r.close()
    r = make(capi, scenes["mixed"], cameras["mixed"], 640, 360, 16)
    r.set_tuning(0, 0, 0, variant)
    r.render(1, 4)
    assert crc(r.read_accum()[:, :, :3]) == meta["radiance_mixed_b16_640x360"]["crc_mean4"]
    r.close()


def test_workspace_batches_and_item_sizes_do_not_change_results(capi, cameras, scenes):
    """the per-pass radiance workspace may be split into batches and passes grouped per work item in any
    way: the image must stay bit-identical"""
    W, H, SPP = 128, 72, 10
    ref = make(capi, scenes["mixed"], cameras["mixed"], W, H, 6)
    ref.render(1, SPP)
    want = ref.read_accum()
    for chunk, limit in ((1, 0), (3, 0), (64, 0), (4, 3 * W * H * 12), (2, 1)):     # 12 B per sample in the workspace: 3 passes per batch -> 4 launches
        r = make(capi, scenes["mixed"], cameras["mixed"], W, H, 6)
        r.set_tuning(chunk, limit, 0, 4)       # brute-force kernel: per-pass workspace + pt_accumulate
        r.render(1, SPP)
        ms, launches = r.last_render_launches()
        assert bits_equal(r.read_accum(), want), (chunk, limit)
        if limit == 1:
            assert launches == SPP
        elif limit:
            assert launches == 4
        r.close()
    ref.close()


def test_bvh_heightfield_8k_triangles_vs_oracle(capi, cameras):
    """config-3-shaped mesh (tessellated height field) small enough for the brute-force oracle: the BVH traversal
    (library default for this size) must reproduce it bit for bit, primary rays and bounced paths"""
    import oracle_py as O
    from path_tracer_amd import scene as S
    m = S.heightfield_mesh(64, 1)                      # 8192 triangles
    sc = S.default_scene(m)
    cam = S.Camera(position=np.array([0.2, 1.6, 2.2], np.float32), yaw=95.0, pitch=35.0).UpdateView()
    W, H = 192, 104
    r = make(capi, sc, cam, W, H, 6)
    r.reset_counters()
    r.render(1, 4)
    c = r.counters()
    assert c["bvh_tri_tests"] > 0 and c["mesh_tri_tests"] == 0, "the BVH kernel did not run"
    assert c["bvh_tri_tests"] < 0.02 * c["mesh_segments"] * len(m.tris), "the BVH culls nothing"
    want = O.Oracle(sc, cam, W, H, 6).render(1, 4)
    assert n_diff_pixels(r.read_accum(), want) == 0
    r.close()


def test_config3_1m_triangle_mesh_1080p_windows_vs_oracle(capi):
    """BASELINE.json configs[2]: ~1M-triangle tessellated height field at 1080p through the BVH traversal.  The
    brute-force oracle costs ~1M ray-triangle tests per segment, so it checks two small windows (bit-exact, bounced
    paths included); the whole frame is checked for determinism and against the brute-force HIP kernel on a strip."""
    import oracle_py as O
    from path_tracer_amd import scene as S
    m = S.heightfield_mesh(708, 3)
    assert len(m.tris) == 1002528
    sc = S.default_scene(m)
    cam = S.Camera(position=np.array([0.2, 1.6, 2.2], np.float32), yaw=95.0, pitch=35.0).UpdateView()
    W, H, B, SPP = 1920, 1080, 4, 2
    r = make(capi, sc, cam, W, H, B)
    r.reset_counters()
    r.render(1, SPP)
    full = r.read_accum()
    c = r.counters()
    assert c["samples"] == W * H * SPP and c["bvh_tri_tests"] > 0 and c["mesh_segments"] > 0.2 * c["samples"]
    assert c["bvh_tri_tests"] < 50 * c["mesh_segments"], "more than 50 of 1M triangles tested per ray"
    r.reset()
    r.render(1, 1)
    r.render(2, 1)
    assert crc(r.read_accum()) == crc(full)
    o = O.Oracle(sc, cam, W, H, B)
    for (x0, y0, w, h) in ((940, 500, 24, 4), (300, 700, 16, 4)):
        want = o.render_region(1, SPP, x0, y0, w, h)
        assert n_diff_pixels(full[y0:y0 + h, x0:x0 + w], want) == 0, (x0, y0)
    r.close()
    # brute-force HIP kernel (1424 register chunks per gate-passing ray) on a small frame with the same camera
    Ws, Hs = 64, 40
    a = make(capi, sc, cam, Ws, Hs, B)
    a.render(1, 1)
    b = make(capi, sc, cam, Ws, Hs, B)
    b.set_tuning(0, 0, 0, 1)
    b.render(1, 1)
    assert bits_equal(a.read_accum(), b.read_accum())
    a.close(); b.close()


def test_config3_at_baseline_size_64spp_8_bounces(capi):
    """BASELINE.json configs[2] at its full size -- 1,002,528 triangles, 1920x1080, 64 spp, 8 bounces -- rendered once:
    sample and segment counters, run-to-run determinism (two renders, equal CRC), and two 24x4 windows compared bit for bit
    with the brute-force oracle (2 x 96 px x 64 spp x ~2.3 segments x 1,002,528 triangle tests, tens of seconds of CPU)."""
    import oracle_py as O
    from path_tracer_amd import scene as S
    sc = S.default_scene(S.heightfield_mesh(708, 3))
    cam = S.heightfield_camera()
    W, H, B, SPP = 1920, 1080, 8, 64
    r = make(capi, sc, cam, W, H, B)
    r.reset_counters()
    r.render(1, SPP)
    full = r.read_accum()
    c = r.counters()
    assert c["samples"] == W * H * SPP and c["segments"] > 2 * c["samples"] and c["bvh_tri_tests"] > 0 and c["mesh_tri_tests"] == 0
    assert c["bvh_tri_tests"] < 50 * c["mesh_segments"], "more than 50 of 1M triangles tested per ray"
    assert r.kernel_name().startswith("pt_trace_items_bvh<")       # the BVH lives in global memory at this size
    r.reset()
    r.render(1, SPP)
    assert crc(r.read_accum()) == crc(full)
    assert np.all(full[:, :, 3] == 1.0) and np.isfinite(full).all()
    r.close()
    o = O.Oracle(sc, cam, W, H, B)
    for (x0, y0, w, h) in ((940, 500, 24, 4), (300, 700, 24, 4)):
        want = o.render_region(1, SPP, x0, y0, w, h)
        assert n_diff_pixels(full[y0:y0 + h, x0:x0 + w], want) == 0, (x0, y0)


@pytest.mark.parametrize("variant", [0, 1, 5, 6])
@pytest.mark.parametrize("name", ["dragon1", "suzanne"])
def test_other_reference_meshes_vs_reference_fixtures(capi, name, variant, meta, cameras, mesh_scenes):
    """two more meshes of the reference's res/meshes against the reference's own shader under llvmpipe: dragon1 (5,438 triangles:
    eight register chunks for the brute-force kernel, a tree too large for LDS -> the quantised four-wide nodes read in place)
    and suzanne in the mixed scene (967 triangles + 145 spheres, trees in LDS); default kernel, brute force, both traversal
    kernels; running means after passes 1, 2, 4, 8 at 128x72, primary-hit ids, and the 640x360 CRC"""
    from conftest import MESH_CASES
    c = MESH_CASES[name]
    sc, cam, b = mesh_scenes[name], cameras[c["cam"]], c["bounces"]
    g = np.load(os.path.join(GOLD, "radiance_meshes_128x72.npz"))
    r = make(capi, sc, cam, 128, 72, b)
    if variant:
        r.set_tuning(0, 0, 0, variant)
    first = 1
    for k in (1, 2, 4, 8):
        r.render(first, k - first + 1)
        first = k + 1
        assert n_diff_pixels(r.read_accum()[:, :, :3], g[f"{name}_b{b}_128x72_mean{k}"]) == 0, f"running mean after pass {k}"
    want = {0: "pt_trace_items_bvh", 1: "pt_trace_items<", 5: "pt_trace_items_bvh<", 6: "pt_trace_items_bvh"}[variant]
    assert r.kernel_name().startswith(want), r.kernel_name()
    if name == "dragon1" and variant in (0, 6):
        assert r.kernel_name().startswith("pt_trace_items_bvh<"), "5,438 triangles do not fit in LDS: the tree is read in place"
    ids = r.read_primary_hits(1)[0]
    assert np.array_equal(ids.astype(np.int16), g[f"{name}_128x72_p1_ids"])
    r.close()
    r = make(capi, sc, cam, 640, 360, b)
    if variant:
        r.set_tuning(0, 0, 0, variant)
    r.render(1, 2)
    assert crc(r.read_accum()[:, :, :3]) == meta[f"radiance_{name}_b{b}_640x360"]["crc_mean2"]
    r.close()


def test_config5_at_baseline_size_64spp_16_bounces(capi, cameras, scenes):
    """BASELINE.json configs[4] on one GPU at its full size -- the mixed scene (bunny1 diffuse + specular, the 12 x 12 grid of
    glass spheres and the large sphere of src/Main.cpp:76-92, emissive and ground triangles), 1920x1080, 64 spp, 16 bounces --
    rendered once: counters, the kernel that ran (sphere tree, trees staged in LDS), run-to-run determinism (two renders, equal
    CRC), and three windows compared bit for bit with the oracle: 128 x 8 on the glass grid (11 spheres under the window's
    first rows), 128 x 8 on the bunny, 64 x 4 across the horizon."""
    import oracle_py as O
    sc, cam = scenes["mixed"], cameras["mixed"]
    W, H, B, SPP = 1920, 1080, 16, 64
    r = make(capi, sc, cam, W, H, B)
    r.reset_counters()
    r.render(1, SPP)
    full = r.read_accum()
    c = r.counters()
    assert c["samples"] == W * H * SPP and c["segments"] > 3 * c["samples"] and c["hits"] > 2 * c["samples"]
    assert c["bvh_sphere_tests"] > c["samples"] and c["bvh_tri_tests"] > 0 and c["mesh_tri_tests"] == 0
    assert c["bvh_sphere_tests"] < 20 * c["segments"], "more than 20 of 145 spheres tested per segment"
    assert r.kernel_name().startswith("pt_trace_items_bvh_lds<true, false, false, ")     # sphere tree, long launch, rt.comp's forms
    r.reset()
    r.render(1, SPP)
    assert crc(r.read_accum()) == crc(full)
    assert np.all(full[:, :, 3] == 1.0) and np.isfinite(full).all()
    r.close()
    o = O.Oracle(sc, cam, W, H, B)
    for (x0, y0, w, h) in ((896, 528, 128, 8), (856, 544, 128, 8), (40, 900, 64, 4)):
        want = o.render_region(1, SPP, x0, y0, w, h)
        assert n_diff_pixels(full[y0:y0 + h, x0:x0 + w], want) == 0, (x0, y0)


@pytest.mark.parametrize("n_tris", [1, 3, 5])
def test_bvh_tiny_meshes(capi, cameras, n_tris):
    """meshes at and below the leaf size (the BVH root is then a stand-in inner node over one leaf)"""
    import oracle_py as O
    from path_tracer_amd import scene as S
    m = S.heightfield_mesh(4, 1)
    tris = m.tris[:n_tris].copy()
    mesh = S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 1)
    sc = S.default_scene(mesh)
    sc.add_sphere((0.3, 0.5, 0.2), 0.25, 2)             # one sphere: the sphere tree is a stand-in root too
    cam = S.Camera(position=np.array([0.2, 1.6, 2.2], np.float32), yaw=95.0, pitch=35.0).UpdateView()
    W, H = 96, 56
    want = O.Oracle(sc, cam, W, H, 5).render(1, 3)
    for variant in (5, 6, 1):
        r = make(capi, sc, cam, W, H, 5)
        r.set_tuning(0, 0, 0, variant)
        r.render(1, 3)
        assert n_diff_pixels(r.read_accum(), want) == 0, variant
        r.close()


def test_bvh_refuses_non_finite_mesh_and_falls_back(capi, cameras):
    """a NaN vertex makes the builder refuse the mesh; the library must then trace it by brute force (same image
    as the oracle, where that triangle simply never hits)"""
    import oracle_py as O
    from path_tracer_amd import scene as S
    m = S.heightfield_mesh(8, 1)
    tris = m.tris.copy()
    tris["p2"][5][1] = np.nan
    mesh = S.Mesh(tris, S.Mesh.gen_bounding_box(m.tris), 1)
    sc = S.default_scene(mesh)
    cam = S.Camera(position=np.array([0.2, 1.6, 2.2], np.float32), yaw=95.0, pitch=35.0).UpdateView()
    W, H = 64, 40
    r = make(capi, sc, cam, W, H, 4)
    r.reset_counters()
    r.render(1, 2)
    c = r.counters()
    assert c["bvh_tri_tests"] == 0 and c["mesh_tri_tests"] > 0, "expected the brute-force kernel"
    want = O.Oracle(sc, cam, W, H, 4).render(1, 2)
    assert n_diff_pixels(r.read_accum(), want) == 0
    r.close()


def test_config4_4k_sample_split_emulated_on_one_gpu(capi, cameras, scenes):
    """BASELINE.json configs[3] (4K, sample split + reduce) with the ranks emulated on one GPU: four contexts trace
    the pass ranges path_tracer_amd.distributed.pass_range gives ranks 0..3 of a 4-GPU job as plain sums into their
    own device buffers (torch tensors, as in bench.py), the buffers are summed the way the RCCL reduce sums them,
    rank 0 scales.  Checked against the oracle's running mean on windows (rtol 1e-5: the summation order differs)."""
    import torch
    import oracle_py as O
    from path_tracer_amd import distributed as D
    W, H, B, SPP, WORLD = 3840, 2160, 4, 16, 4
    sc, cam = scenes["default"], cameras["bench"]
    total = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    covered = []
    for rank in range(WORLD):
        first, count = D.pass_range(rank, WORLD, SPP)
        covered += list(range(first, first + count))
        buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        r = make(capi, sc, cam, W, H, B)
        r.set_accum_buffer(buf.data_ptr(), buf.numel() * 4)
        r.render_sum(first, count)
        r.synchronize()
        total += buf                                   # what dist.reduce(SUM) does across ranks
        r.close()
    assert covered == list(range(1, SPP + 1))
    assert D.reduce_and_finish(total, SPP)             # world size 1: no collective, just the finish on "rank 0"
    got = total.cpu().numpy()
    assert np.all(got[:, :, 3] == 1.0) and np.isfinite(got).all()
    o = O.Oracle(sc, cam, W, H, B)
    for (x0, y0, w, h) in ((1800, 760, 96, 6), (0, 0, 64, 4), (3776, 2156, 64, 4), (2000, 600, 64, 6)):
        want = o.render_region(1, SPP, x0, y0, w, h)
        np.testing.assert_allclose(got[y0:y0 + h, x0:x0 + w, :3], want[:, :, :3], rtol=1e-5, atol=1e-6)


def test_fused_accumulation_equals_workspace_path(capi, cameras, scenes):
    """the traversal kernels keep the running mean in registers (one lane = all passes of a pixel); with option no_fused = 1
    they write per-pass radiance and pt_accumulate folds it: same image bit for bit, also when a render continues an
    earlier one (first_pass > 1) and in sum mode"""
    W, H = 160, 88
    imgs = {}
    for no_fused in ("0", "1"):
        r = make(capi, scenes["mixed"], cameras["mixed"], W, H, 6)
        r.set_option("no_fused", int(no_fused))
        r.render(1, 5)
        r.render(6, 3)
        a = r.read_accum()
        r.reset()
        r.render_sum(2, 4)
        r.render_sum(6, 2)
        imgs[no_fused] = (a, r.read_accum(), r.last_render_launches()[1])
        r.close()
    assert bits_equal(imgs["0"][0], imgs["1"][0]) and bits_equal(imgs["0"][1], imgs["1"][1])
    assert np.all(imgs["0"][1][:, :, 3] == 6.0)


def _octahedron(centre, radius, material):
    from path_tracer_amd import scene as S
    c = np.asarray(centre, np.float32)
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32) * np.float32(radius) + c
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int32)
    return S.Mesh.from_arrays(v, f, material), v, f


@pytest.mark.parametrize("variant", [1, 6])
def test_merged_meshes_with_their_own_materials_vs_oracle(capi, cameras, variant):
    """several meshes traced as one (the reference keeps only the mesh built last): bunny (ground material)
    + a mirror-like octahedron (material 0) + a glass one (material 2), per-triangle materials, one root AABB"""
    import oracle_py as O
    from path_tracer_amd import scene as S
    parts = [S.bunny_mesh(3), _octahedron((0.3, 0.35, 0.7), 0.25, 0)[0], _octahedron((0.2, 0.3, -0.7), 0.2, 2)[0]]
    m = S.Mesh.merged(parts)
    assert len(m.tris) == len(parts[0].tris) + 16 and m.tris["materialIndex"][-1] == 2 and m.tris["materialIndex"][0] == 3
    sc = S.default_scene(m)
    W, H = 160, 88
    r = make(capi, sc, cameras["bench"], W, H, 6)
    r.set_tuning(0, 0, 0, variant)
    r.render(1, 3)
    got = r.read_accum()
    want = O.Oracle(sc, cameras["bench"], W, H, 6).render(1, 3)
    assert n_diff_pixels(got, want) == 0
    alone = make(capi, S.default_scene(parts[0]), cameras["bench"], W, H, 6)
    alone.render(1, 3)
    assert n_diff_pixels(alone.read_accum(), want) > 100, "the extra meshes are not in view"
    alone.close()
    r.close()


@pytest.mark.parametrize("size", [(8, 8), (72, 40), (100, 61), (200, 95), (333, 17)])
@pytest.mark.parametrize("spp", [2, 6])
def test_ragged_image_sizes_vs_oracle(capi, cameras, scenes, size, spp):
    """widths that are not a multiple of the 64-pixel tile (partly filled last tile), sizes that are not a multiple of the 8x8
    dispatch group (untouched border), the 8x8 minimum; 2 spp takes the short-launch kernels, 6 spp the workspace path"""
    import oracle_py as O
    W, H = size
    r = make(capi, scenes["mixed"], cameras["mixed"], W, H, 5)
    r.render(1, spp)
    got = r.read_accum()
    want = O.Oracle(scenes["mixed"], cameras["mixed"], W, H, 5).render(1, spp)
    tw, th = (W // 8) * 8, (H // 8) * 8
    assert n_diff_pixels(got[:th, :tw], want[:th, :tw]) == 0
    assert np.all(got[th:, :, :] == 0) and np.all(got[:, tw:, :] == 0)
    r.close()



def test_one_shared_lds_copy_per_cu_for_trees_that_do_not_fit_twice(capi, cameras, scenes):
    """pt_trace_items_bvh_lds<..., 1024>: one 1024-thread workgroup per CU shares one LDS copy of the trees where two copies
    (two 512-thread workgroups) do not fit -- the mixed scene's bunny + 145 spheres; forced on the default scene it must give
    the image of the 512-thread kernel bit for bit"""
    g = np.load(os.path.join(GOLD, "radiance_128x72.npz"))
    r = make(capi, scenes["mixed"], cameras["mixed"], 128, 72, 16)
    r.render(1, 16)
    assert r.kernel_name().startswith("pt_trace_items_bvh_lds<") and r.kernel_name().endswith(", 1024>"), r.kernel_name()
    assert n_diff_pixels(r.read_accum()[:, :, :3], g["mixed_b16_128x72_mean16"]) == 0
    r.close()
    # a mesh small enough for either shape (200 triangles): both forced, the images must agree bit for bit
    from path_tracer_amd import scene as S
    sc, cam = S.default_scene(S.heightfield_mesh(10, 3)), S.heightfield_camera()
    imgs = {}
    for block in (512, 1024):
        r = make(capi, sc, cam, 640, 360, 8)
        r.set_option("lds_block", block)
        r.render(1, 6)
        assert r.kernel_name().endswith(f", {block}>"), r.kernel_name()
        imgs[block] = r.read_accum()
        r.close()
    assert bits_equal(imgs[512], imgs[1024])
    # a forced shape the trees do not fit falls back to the kernel that reads them in place (variant 5), same image
    r = make(capi, scenes["mixed"], cameras["mixed"], 128, 72, 16)
    r.set_option("lds_block", 512)
    r.render(1, 16)
    assert r.kernel_name().startswith("pt_trace_items_bvh<true"), r.kernel_name()
    assert n_diff_pixels(r.read_accum()[:, :, :3], g["mixed_b16_128x72_mean16"]) == 0
    r.close()
