import json
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def n_diff_pixels(a, b):
    """pixels whose float bits differ in any channel (a, b: (H,W,C))"""
    return int((np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)).any(-1).sum())


@pytest.fixture(scope="session")
def meta():
    return json.load(open(os.path.join(GOLD, "meta.json")))["entries"]


@pytest.fixture(scope="session")
def cameras():
    """Camera matrices produced by the reference's own GLM (tests/golden/camera.json)."""
    from path_tracer_amd import scene as S
    out = {}
    names = ["origin", "bench", "mixed", "rot", "cam4", "cam5"]
    for name, e in zip(names, json.load(open(os.path.join(GOLD, "camera.json")))["cameras"]):
        c = S.Camera(position=np.array([int(b, 16) for b in e["position_bits"]], np.uint32).view(np.float32), yaw=e["yaw"], pitch=e["pitch"])
        c.inverseView = np.array([int(b, 16) for b in e["inverse_view_bits"]], np.uint32).view(np.float32).reshape(4, 4)
        out[name] = c
    return out


@pytest.fixture(scope="session")
def scenes():
    from path_tracer_amd import scene as S
    return {"default": S.default_scene(), "mixed": S.mixed_scene()}


SCENE_CAM = {"default": ("default", "bench"), "default_rot": ("default", "rot"), "mixed": ("mixed", "mixed")}


MESH_CASES = {"dragon1": dict(material=3, mixed=False, cam="bench", bounces=8), "suzanne": dict(material=1, mixed=True, cam="mixed", bounces=16)}


@pytest.fixture(scope="session")
def mesh_scenes():
    """other meshes of the reference's res/meshes (parsed arrays in tests/golden/mesh_*.npz: data) in main()'s scene / the mixed
    scene, as oracle/gen_goldens.py rendered them with the reference's rt.comp (radiance_meshes_128x72.npz)"""
    from path_tracer_amd import scene as S
    out = {}
    for name, c in MESH_CASES.items():
        z = np.load(os.path.join(GOLD, f"mesh_{name}.npz"))
        mesh = S.Mesh.from_arrays(z["vertices"], z["faces"], c["material"])
        out[name] = S.mixed_scene(mesh) if c["mixed"] else S.default_scene(mesh)
    return out
