"""GPU (-m gpu): pt_set_shading_params -- the literals in which the reference's legacy res/shaders/rt.frag differs from
rt.comp, as arguments of the same kernels.

The rt.frag preset is checked against fixtures of rt.frag itself run under llvmpipe (tests/golden/radiance_frag_128x72.npz,
generator oracle/gen_goldens.py --only frag); single-field variations, which no shader of the reference has, against the
CPU oracle.  Everything bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLD, SCENE_CAM, crc, n_diff_pixels
import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as m
    m.load()
    return m


def make(capi, sc, cam, W, H, bounces, debug_normal=False, preset=None, **fields):
    r = capi.Renderer(W, H, device=0)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    r.SetUniform1i("debugNormal", int(debug_normal))
    if preset is not None or fields:
        r.set_shading_params(preset, **fields)
    return r


def _gen(kernel_name):
    """third template argument of the traversal kernels: <SPH, SHORT, GEN[, BLOCK]>"""
    return kernel_name.split("<")[1].rstrip(">").split(",")[2].strip()


@pytest.mark.parametrize("tag,bounces", [("default", 12), ("default", 4), ("mixed", 12), ("mixed", 4)])
def test_frag_preset_running_mean_vs_rt_frag_fixtures(capi, tag, bounces, cameras, scenes):
    """one launch per pass, like the legacy frame loop: the running mean after passes 1, 2, 3, 4 and 16"""
    g = np.load(os.path.join(GOLD, "radiance_frag_128x72.npz"))
    r = make(capi, scenes[SCENE_CAM[tag][0]], cameras[SCENE_CAM[tag][1]], 128, 72, bounces, preset=capi.RT_FRAG)
    for k in range(1, 17):
        r.render_pass(k)
        if k in (1, 2, 3, 4, 16):
            acc = r.read_accum()
            assert n_diff_pixels(acc[:, :, :3], g[f"{tag}_b{bounces}_128x72_mean{k}"]) == 0, f"running mean after pass {k}"
            assert np.all(acc[:, :, 3] == 1.0)
    assert _gen(r.kernel_name()) == "true", r.kernel_name()           # the instantiation that reads the form flags
    r.close()


@pytest.mark.parametrize("tag", ["default", "mixed"])
def test_frag_preset_batched_launches_and_normals(capi, tag, meta, cameras, scenes):
    g = np.load(os.path.join(GOLD, "radiance_frag_128x72.npz"))
    sc, cam = scenes[SCENE_CAM[tag][0]], cameras[SCENE_CAM[tag][1]]
    r = make(capi, sc, cam, 128, 72, 12, preset=capi.RT_FRAG)
    r.render(1, 16)                                    # workspace + pt_accumulate
    assert n_diff_pixels(r.read_accum()[:, :, :3], g[f"{tag}_b12_128x72_mean16"]) == 0
    r.reset()
    r.render(1, 4)                                     # few passes: the running mean stays in registers
    assert n_diff_pixels(r.read_accum()[:, :, :3], g[f"{tag}_b12_128x72_mean4"]) == 0
    r.reset()
    r.render_static(1, 3)                              # the entry point that allocates nothing shares the kernels
    assert n_diff_pixels(r.read_accum()[:, :, :3], g[f"{tag}_b12_128x72_mean3"]) == 0
    r.close()
    r = make(capi, sc, cam, 128, 72, 12, debug_normal=True, preset=capi.RT_FRAG)
    r.render_pass(1)
    assert n_diff_pixels(r.read_accum()[:, :, :3], g[f"{tag}_128x72_normal"]) == 0
    r.close()
    r = make(capi, sc, cam, 640, 360, 12, preset=capi.RT_FRAG)
    r.render(1, 4)
    assert crc(r.read_accum()[:, :, :3]) == meta[f"radiance_frag_{tag}_b12_640x360"]["crc_mean4"]
    r.close()


@pytest.mark.parametrize("variant", [1, 2, 5, 6])
def test_frag_preset_in_every_kernel_family(capi, variant, cameras, scenes):
    """brute force in registers / in LDS, traversal with the BVH in global memory / in LDS"""
    g = np.load(os.path.join(GOLD, "radiance_frag_128x72.npz"))
    r = make(capi, scenes["mixed"], cameras["mixed"], 128, 72, 12, preset=capi.RT_FRAG)
    r.set_tuning(variant=variant)
    r.render(1, 2)
    assert n_diff_pixels(r.read_accum()[:, :, :3], g["mixed_b12_128x72_mean2"]) == 0
    r.close()


FIELDS = [("hit_limit", 0.0), ("hit_limit", 5e-4), ("origin_offset", 0.00015), ("jitter_scale", 0.0005), ("jitter_scale", 0.0),
          ("sky_color", (0.92, 0.8, 0.85)), ("sky_intensity", 0.8), ("fresnel_n1_one", 1), ("two_hemisphere_draws", 1),
          ("pixel_center", 1), ("sqrt_output", 1), ("trace_mesh", 0), ("sphere_sum_xyz", 1)]


@pytest.mark.parametrize("field,value", FIELDS, ids=[f"{k}={v}" for k, v in FIELDS])
def test_single_field_variations_vs_oracle(capi, field, value, cameras, scenes):
    """rt.comp with ONE literal changed: no shader of the reference has these, the CPU oracle is the only check"""
    sc, cam = scenes["mixed"], cameras["mixed"]
    want = O.Oracle(sc, cam, 128, 72, 8, preset=0, params={field: value}).render(1, 3)
    base = O.Oracle(sc, cam, 128, 72, 8).render(1, 3)
    r = make(capi, sc, cam, 128, 72, 8, **{field: value})
    r.render(1, 3)
    got = r.read_accum()
    assert n_diff_pixels(got, want) == 0
    if field not in ("hit_limit",):                    # the hit limit shows in a handful of pixels of larger images only
        assert n_diff_pixels(want, base) > 0, "the variation must change the image, or this test checks nothing"
    r.close()


def test_frag_preset_with_mesh_and_rt_comp_forms_vs_oracle(capi, cameras, scenes):
    """mixed presets: rt.frag's literals on rt.comp's mesh path, 1920x1080 window (traversal with parked rays, sphere tree)"""
    sc, cam = scenes["mixed"], cameras["mixed"]
    W, H = 1920, 1080
    r = make(capi, sc, cam, W, H, 12, preset=capi.RT_FRAG, trace_mesh=1)
    r.render(1, 2)
    got = r.read_accum()
    o = O.Oracle(sc, cam, W, H, 12, preset=1, params={"trace_mesh": 1})
    x0, y0, w, h = 700, 380, 512, 96
    want = o.render_region(1, 2, x0, y0, w, h)
    assert n_diff_pixels(got[y0:y0 + h, x0:x0 + w], want) == 0
    r.close()


def test_api_contract(capi, cameras, scenes):
    p = capi.shading_preset(capi.RT_COMP).as_dict()
    assert p["hit_limit"] == np.float32(0.00001) and p["origin_offset"] == np.float32(0.0002) and p["jitter_scale"] == np.float32(0.00025)
    assert p["sky_color"] == (1.0, 1.0, 1.0) and p["sky_intensity"] == 0.5 and p["trace_mesh"] == 1
    assert [p[k] for k in ("fresnel_n1_one", "two_hemisphere_draws", "pixel_center", "sqrt_output", "sphere_sum_xyz")] == [0] * 5
    q = capi.shading_preset(capi.RT_FRAG).as_dict()
    assert q["hit_limit"] == 0.0 and q["origin_offset"] == np.float32(0.00015) and q["jitter_scale"] == np.float32(0.0005)
    assert q["sky_color"] == tuple(float(np.float32(v)) for v in (0.92, 0.8, 0.85)) and q["sky_intensity"] == np.float32(0.8) and q["trace_mesh"] == 0
    with pytest.raises(ValueError):
        capi.shading_preset(2)
    sc, cam = scenes["mixed"], cameras["mixed"]
    r = make(capi, sc, cam, 128, 72, 3)
    assert r.shading_params().as_dict() == p           # a new context has rt.comp's values
    for bad in ({"hit_limit": -1e-6}, {"hit_limit": 0.01}, {"origin_offset": -1.0}, {"jitter_scale": float("nan")},
                {"sky_color": (1.0, float("inf"), 1.0)}, {"sqrt_output": 2}, {"trace_mesh": -1}):
        with pytest.raises(RuntimeError):
            r.set_shading_params(**bad)
        assert r.shading_params().as_dict() == p       # a rejected call changes nothing
    # switching presets back and forth on one context (the Fresnel table follows): rt.comp's own fixture again
    g = np.load(os.path.join(GOLD, "radiance_128x72.npz"))
    r.set_shading_params(capi.RT_FRAG)
    assert r.shading_params().as_dict() == q
    r.render(1, 2)
    r.reset()
    r.set_shading_params(capi.RT_COMP)
    r.render(1, 2)
    assert n_diff_pixels(r.read_accum()[:, :, :3], g["mixed_b3_128x72_mean2"]) == 0
    assert _gen(r.kernel_name()) == "false", r.kernel_name()
    r.close()
