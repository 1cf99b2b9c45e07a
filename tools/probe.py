#!/usr/bin/env python3
"""tools/probe.py -- time one configuration of the hot path on the GPU and print its counters (perf experiments).
   python tools/probe.py --scene default|mixed|empty --cam x y z yaw pitch --size W H --spp N --bounces B --chunk C --variant V"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from path_tracer_amd import capi, scene as S

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="default")
ap.add_argument("--cam", type=float, nargs=5, default=[1.5, 0.4, 0.0, 0.0, 0.0])
ap.add_argument("--size", type=int, nargs=2, default=[1920, 1080])
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--waves", type=int, default=0)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="pt_set_option (repeatable), e.g. --opt park_high=48")
a = ap.parse_args()
import time
if a.scene.startswith("hf"):
    t0 = time.time(); sc = S.default_scene(S.heightfield_mesh(int(a.scene[2:]), 3)); print("mesh built in %.1f s, %d triangles" % (time.time() - t0, len(sc.mesh_tris)), flush=True)
else:
    sc = {"default": S.default_scene, "mixed": S.mixed_scene}[a.scene]()
cam = S.Camera(position=a.cam[:3], yaw=a.cam[3], pitch=a.cam[4]); cam.UpdateView()
r = capi.Renderer(a.size[0], a.size[1])
t0 = time.time(); r.upload_scene(sc); print("upload (incl. BVH build) %.2f s" % (time.time() - t0), flush=True); r.SetUniformCamera(cam); r.SetUniform1i("maxBounces", a.bounces)
if a.chunk or a.waves or a.variant:
    r.set_tuning(a.chunk, 0, a.waves, a.variant)
for kv in a.opt:
    k, v = kv.split("="); r.set_option(k, int(v))
ms = []
for i in range(a.reps + 1):
    r.reset(); r.reset_counters(); r.render(1, a.spp); ms.append(r.last_render_ms())
c = r.counters()
bal = c["wave_ticks_sum"] / max(1, c["waves"] * c["wave_ticks_max"])
import zlib
print(json.dumps({"crc": zlib.crc32(r.read_accum().tobytes()), "args": vars(a), "kernel_ms": [round(m, 3) for m in ms[1:]], "balance": round(bal, 3),
                  "span_max_ms": c["wave_ticks_max"] / 1e5, **{k: c[k] for k in ("samples", "segments", "mesh_segments", "hits", "wave_iterations", "bvh_node_visits", "bvh_tri_tests", "mesh_tri_tests", "phase_cycles")}}))
