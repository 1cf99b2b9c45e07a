"""ctypes binding of oracle/_build/libpt_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package path_tracer_amd never does.  See oracle/pt_oracle.h for the parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpt_oracle.so")


class OrScene(C.Structure):
    _fields_ = [("mesh_tris", C.c_void_p), ("n_mesh_tris", C.c_int32), ("nodes", C.c_void_p), ("n_nodes", C.c_int32),
                ("materials", C.c_void_p), ("n_materials", C.c_int32), ("spheres", C.c_void_p), ("n_spheres", C.c_int32),
                ("tris", C.c_void_p), ("n_tris", C.c_int32)]


class OrParams(C.Structure):
    _fields_ = [("hit_limit", C.c_float), ("origin_offset", C.c_float), ("jitter_scale", C.c_float), ("sky_color", C.c_float * 3),
                ("sky_intensity", C.c_float), ("fresnel_n1_one", C.c_int32), ("two_hemisphere_draws", C.c_int32),
                ("pixel_center", C.c_int32), ("sqrt_output", C.c_int32), ("trace_mesh", C.c_int32), ("sphere_sum_xyz", C.c_int32)]


class OrFrame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("cam_pos", C.c_float * 3), ("inv_view", C.c_float * 16),
                ("max_bounces", C.c_int32), ("debug_normal", C.c_int32), ("prm", OrParams)]


class OrBounce(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("prim_id", C.c_int32), ("t", C.c_float), ("rng_state", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc) if the .so is missing or stale."""
    srcs = [os.path.join(_HERE, n) for n in ("pt_oracle.c", "pt_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "_build/libpt_oracle.so"], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.or_next_random.restype = C.c_uint32
        L.or_next_random.argtypes = [C.POINTER(C.c_uint32)]
        L.or_random_value.restype = C.c_float
        L.or_random_value.argtypes = [C.POINTER(C.c_uint32)]
        for fn in (L.or_logf, L.or_cosf):
            fn.restype, fn.argtypes = C.c_float, [C.c_float]
        L.or_random_value_nd.restype = C.c_float
        L.or_random_value_nd.argtypes = [C.POINTER(C.c_uint32)]
        L.or_pixel_index.restype = C.c_uint32
        L.or_pixel_index.argtypes = [C.c_int32] * 4
        L.or_render.argtypes = [C.POINTER(OrScene), C.POINTER(OrFrame), C.c_int32, C.c_int32, C.c_void_p]
        L.or_render_pass.argtypes = [C.POINTER(OrScene), C.POINTER(OrFrame), C.c_int32, C.c_void_p]
        L.or_radiance_pass.argtypes = [C.POINTER(OrScene), C.POINTER(OrFrame), C.c_int32, C.c_void_p]
        L.or_primary_hits.argtypes = [C.POINTER(OrScene), C.POINTER(OrFrame), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_count_work.argtypes = [C.POINTER(OrScene), C.POINTER(OrFrame), C.c_int32, C.c_int32, C.c_void_p]
        L.or_ray_trace.argtypes = [C.POINTER(OrScene), C.POINTER(OrFrame), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p,
                                   C.c_void_p, C.POINTER(C.c_int32)]
        L.or_primary_ray.restype = C.c_uint32
        L.or_primary_ray.argtypes = [C.POINTER(OrFrame), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.or_render_region.argtypes = [C.POINTER(OrScene), C.POINTER(OrFrame)] + [C.c_int32] * 6 + [C.c_void_p]
        L.or_num_threads.restype = C.c_int32
        L.or_params_preset.argtypes = [C.POINTER(OrParams), C.c_int32]
    return _lib


class Oracle:
    """Holds references to the scene arrays (so the pointers stay valid) and a frame description."""

    def __init__(self, scene, camera, width, height, max_bounces, debug_normal=False, preset=0, params=None):
        """preset: 0 = res/shaders/rt.comp (the live kernel), 1 = the legacy res/shaders/rt.frag constant set; params: dict
        of OrParams fields that override the preset"""
        self._keep = [np.ascontiguousarray(a) for a in (scene.mesh_tris, scene.nodes, scene.materials, scene.spheres, scene.triangles)]
        mt, nd, ma, sp, tr = self._keep
        self.scene = OrScene(mt.ctypes.data, len(mt), nd.ctypes.data, len(nd), ma.ctypes.data, len(ma), sp.ctypes.data, len(sp),
                             tr.ctypes.data, len(tr))
        self.frame = OrFrame()
        self.frame.width, self.frame.height = width, height
        self.frame.cam_pos[:] = [float(v) for v in np.asarray(camera.position, np.float32)]
        self.frame.inv_view[:] = [float(v) for v in camera.inv_view_flat()]
        self.frame.max_bounces, self.frame.debug_normal = max_bounces, int(bool(debug_normal))
        lib().or_params_preset(C.byref(self.frame.prm), preset)
        for k, v in (params or {}).items():
            if k == "sky_color":
                self.frame.prm.sky_color[:] = [float(c) for c in v]
            else:
                setattr(self.frame.prm, k, v)
        self.W, self.H = width, height

    def render(self, first_pass, n_passes, accum=None):
        if accum is None:
            accum = np.zeros((self.H, self.W, 4), np.float32)
        lib().or_render(C.byref(self.scene), C.byref(self.frame), first_pass, n_passes, accum.ctypes.data)
        return accum

    def render_region(self, first_pass, n_passes, x0, y0, w, h):
        out = np.zeros((h, w, 4), np.float32)
        lib().or_render_region(C.byref(self.scene), C.byref(self.frame), first_pass, n_passes, x0, y0, w, h, out.ctypes.data)
        return out

    def radiance_pass(self, p):
        rgb = np.zeros((self.H, self.W, 3), np.float32)
        lib().or_radiance_pass(C.byref(self.scene), C.byref(self.frame), p, rgb.ctypes.data)
        return rgb

    def primary_hits(self, p):
        ids = np.zeros((self.H, self.W), np.int32)
        t = np.zeros((self.H, self.W), np.float32)
        dirs = np.zeros((self.H, self.W, 3), np.float32)
        st = np.zeros((self.H, self.W), np.uint32)
        lib().or_primary_hits(C.byref(self.scene), C.byref(self.frame), p, ids.ctypes.data, t.ctypes.data, dirs.ctypes.data, st.ctypes.data)
        return ids, t, dirs, st

    def count_work(self, first_pass, n_passes):
        c = np.zeros(4, np.uint64)
        lib().or_count_work(C.byref(self.scene), C.byref(self.frame), first_pass, n_passes, c.ctypes.data)
        return {"segments": int(c[0]), "mesh_gate_segments": int(c[1]), "hits": int(c[2]), "samples": int(c[3])}

    def trace_pixel(self, x, y, p):
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        st = C.c_uint32(lib().or_primary_ray(C.byref(self.frame), x, y, p, o.ctypes.data, d.ctypes.data))
        rgb = np.zeros(3, np.float32)
        tr = (OrBounce * max(1, self.frame.max_bounces))()
        n = C.c_int32(0)
        lib().or_ray_trace(C.byref(self.scene), C.byref(self.frame), o.ctypes.data, d.ctypes.data, C.byref(st), rgb.ctypes.data, tr, C.byref(n))
        return rgb, [(tuple(b.origin), tuple(b.dir), b.prim_id, b.t, b.rng_state) for b in tr[: n.value]]


def num_threads() -> int:
    return int(lib().or_num_threads())
