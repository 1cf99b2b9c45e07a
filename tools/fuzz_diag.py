#!/usr/bin/env python3
"""Diagnose one seed of tools/fuzz_bvh.py: which pixel / pass / primitive differs between brute force and the traversal."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_bvh.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from path_tracer_amd import capi
seed, W, H, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sc, cam, bounces = fz.random_scene(seed)
capi.load()
def mk(variant):
    r = capi.Renderer(W, H); r.upload_scene(sc); r.SetUniformCamera(cam); r.SetUniform1i("maxBounces", bounces); r.set_tuning(0, 0, 0, variant); return r
a, b = mk(1), mk(5)
for p in range(1, spp + 1):
    ia, ta, da, _ = a.read_primary_hits(p)
    ib, tb, db, _ = b.read_primary_hits(p)
    bad = np.argwhere((ia != ib) | (ta.view(np.uint32) != tb.view(np.uint32)))
    for (y, x) in bad[:5]:
        i = int(ia[y, x])
        print("pass", p, "pixel", x, y, "brute id/t", i, float(ta[y, x]), "bvh id/t", int(ib[y, x]), float(tb[y, x]), "dir", da[y, x])
        if 0 <= i < len(sc.mesh_tris):
            t = sc.mesh_tris[i]; print("   tri", t["p1"][:3], t["p2"][:3], t["p3"][:3])
print("camera", cam.position, "n_mesh", len(sc.mesh_tris))
