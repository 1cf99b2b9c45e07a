"""CPU: host-side mirror of the reference's Object/Camera surface (path_tracer_amd/scene.py)."""
import json
import os

import numpy as np

from conftest import GOLD
from path_tracer_amd import scene as S


def test_record_layouts_match_std430():
    assert S.MATERIAL_DTYPE.itemsize == 80 and S.MATERIAL_DTYPE.fields["smoothness"][1] == 48 and S.MATERIAL_DTYPE.fields["specularChance"][1] == 68
    assert S.SPHERE_DTYPE.itemsize == 32 and S.SPHERE_DTYPE.fields["radius"][1] == 12 and S.SPHERE_DTYPE.fields["materialIndex"][1] == 16
    assert S.TRIANGLE_DTYPE.itemsize == 64 and S.TRIANGLE_DTYPE.fields["p3"][1] == 32 and S.TRIANGLE_DTYPE.fields["materialIndex"][1] == 48
    assert S.NODE_DTYPE.itemsize == 48 and S.NODE_DTYPE.fields["triIndex"][1] == 32 and S.NODE_DTYPE.fields["childrenIndex"][1] == 40


def test_camera_update_view_matches_reference_glm():
    """UpdateView (src/Camera.cpp:13-26) against matrices computed by the reference's vendored GLM."""
    for e in json.load(open(os.path.join(GOLD, "camera.json")))["cameras"]:
        ref = np.array([int(b, 16) for b in e["inverse_view_bits"]], np.uint32)
        cam = S.Camera(position=np.array(e["position"], np.float32), yaw=e["yaw"], pitch=e["pitch"]).UpdateView()
        assert np.array_equal(cam.inv_view_flat().view(np.uint32), ref), (e["yaw"], e["pitch"])


def test_default_camera_axes():
    cam = S.Camera().UpdateView()
    assert np.allclose(cam.forward, [1, 0, 0]) and np.allclose(cam.right, [0, 0, -1]) and np.allclose(cam.up, [0, 1, 0])
    iv = cam.inverseView
    assert np.allclose(iv[0][:3], [0, 0, 1]) and np.allclose(iv[1][:3], [0, 1, 0]) and np.allclose(iv[2][:3], [-1, 0, 0])


def test_default_scene_contents():
    sc = S.default_scene()
    assert len(sc.mesh_tris) == 656 and len(sc.triangles) == 2 and len(sc.spheres) == 0 and len(sc.materials) == 5
    assert sc.nodes["numTris"][0] == 656 and sc.nodes["childrenIndex"][0] == 1
    assert np.allclose(sc.nodes["boundsMin"][0][:3], [-0.2447, -0.0031, -0.2174], atol=1e-4)
    assert np.allclose(sc.nodes["boundsMax"][0][:3], [0.2517, 0.4870, 0.1696], atol=1e-4)
    assert sc.nodes["boundsMin"][0][3] == 0 and sc.nodes["boundsMax"][0][3] == 0      # w grows over w = 0 (src/Object.cpp:77-87)
    assert (sc.mesh_tris["materialIndex"] == 3).all() and list(sc.triangles["materialIndex"]) == [4, 3]
    assert sc.materials["ior"][2] == np.float32(1.05) and sc.materials["emissionStrength"][4] == 5.0


def test_mixed_scene_contents():
    sc = S.mixed_scene()
    assert len(sc.spheres) == 145 and (sc.mesh_tris["materialIndex"] == 1).all()
    assert sc.spheres["position"][0][0] == np.float32(-2.0) and sc.spheres["position"][0][1] == np.float32(0.101)


def test_obj_loader_roundtrip(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# c\no x\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\ns 0\nf 1 2 3\nf 1/1/1 3/2/2 4/3/3\n")
    v, f = S.load_obj(str(p))
    assert v.shape == (4, 3) and f.tolist() == [[0, 1, 2], [0, 2, 3]]
    m = S.Mesh.from_arrays(v, f, 2)
    assert len(m.tris) == 2 and m.tris["p2"][0][0] == 1.0 and m.tris["p1"][0][3] == 0.0


def test_heightfield_mesh_size():
    m = S.heightfield_mesh(16)
    assert len(m.tris) == 512


def test_merged_mesh_keeps_order_materials_and_one_root_box():
    from path_tracer_amd import scene as S
    a = S.Mesh.from_arrays(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]]), 3)
    b = S.Mesh.from_arrays(np.array([[2, 2, 2], [3, 2, 2], [2, 5, -1]], np.float32), np.array([[0, 1, 2], [2, 1, 0]]), 1)
    m = S.Mesh.merged([a, b])
    assert list(m.tris["materialIndex"]) == [3, 1, 1] and len(m.nodes) == 1 and int(m.nodes["numTris"][0]) == 3
    assert np.array_equal(m.nodes["boundsMin"][0][:3], np.array([0, 0, -1], np.float32))
    assert np.array_equal(m.nodes["boundsMax"][0][:3], np.array([3, 5, 2], np.float32))
    assert np.array_equal(m.tris[1:], b.tris) and np.array_equal(m.tris[:1], a.tris)
    assert len(S.Mesh.merged([]).tris) == 0
