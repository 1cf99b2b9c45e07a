"""ctypes binding of libpt_hip.so (the C ABI declared in include/pt_hip.h).

This is plumbing for tests and bench.py; the product is the shared library.  Importing this module
never touches the GPU; `load()` fails loudly if the library was not built, and `pt_create` fails if no
HIP device is usable -- there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_HIP_LIB") or os.path.join(_HERE, "libpt_hip.so")   # PT_HIP_LIB: A/B builds of the same ABI

PT_OK, PT_E_INVALID, PT_E_HIP, PT_E_STATE, PT_E_LIMIT = 0, -1, -2, -3, -4

# every symbol include/pt_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_resize", "pt_set_accum_buffer", "pt_set_stream", "pt_upload_mesh",
    "pt_upload_scene", "pt_set_camera", "pt_set_debug_normal", "pt_set_max_bounces", "pt_render_pass", "pt_render", "pt_render_static",
    "pt_render_sum", "pt_scale_accum", "pt_reset", "pt_synchronize", "pt_read_accum", "pt_accum_device_ptr",
    "pt_read_primary_hits", "pt_get_counters", "pt_reset_counters", "pt_last_render_ms", "pt_set_tuning", "pt_set_option",
    "pt_shading_preset", "pt_set_shading_params", "pt_get_shading_params", "pt_bvh_selftest", "pt_classify_mesh", "pt_group_unique_id", "pt_group_init", "pt_group_render", "pt_group_finalize",
    "pt_kernel_name", "pt_last_kernel_name",
    "pt_version",
]


class PtShadingParams(C.Structure):
    """include/pt_hip.h PtShadingParams: the literals in which res/shaders/rt.frag differs from rt.comp"""
    _fields_ = [("hit_limit", C.c_float), ("origin_offset", C.c_float), ("jitter_scale", C.c_float), ("sky_color", C.c_float * 3),
                ("sky_intensity", C.c_float), ("fresnel_n1_one", C.c_int32), ("two_hemisphere_draws", C.c_int32),
                ("pixel_center", C.c_int32), ("sqrt_output", C.c_int32), ("trace_mesh", C.c_int32), ("sphere_sum_xyz", C.c_int32)]

    def as_dict(self):
        return {n: (tuple(getattr(self, n)) if n == "sky_color" else getattr(self, n)) for n, _ in self._fields_}


RT_COMP, RT_FRAG = 0, 1      # pt_shading_preset


def shading_preset(preset: int) -> PtShadingParams:
    p = PtShadingParams()
    if load().pt_shading_preset(C.byref(p), int(preset)) != 0:
        raise ValueError(f"no shading preset {preset}")
    return p


class PtCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("mesh_segments", C.c_uint64), ("mesh_tri_tests", C.c_uint64), ("hits", C.c_uint64), ("wave_iterations", C.c_uint64),
                ("primary_mesh_segments", C.c_uint64), ("waves", C.c_uint64), ("wave_ticks_sum", C.c_uint64), ("wave_ticks_max", C.c_uint64),
                ("bvh_node_visits", C.c_uint64), ("bvh_tri_tests", C.c_uint64), ("phase_cycles", C.c_uint64 * 4),
                ("bvh_sphere_tests", C.c_uint64), ("bvh_box_tests", C.c_uint64)]


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pt_hip error {code}: {msg}")
        self.code = code


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so (torch/lib), and two HIP runtimes
    in one process cannot both open the GPU: if libpt_hip.so pulled in /opt/rocm's copy first, a later `import torch`
    brought the second one and whichever touched the card second reported "no HIP device" (seen on MI355X, ROCm 7.2 +
    torch 2.10: tools/hip_order_diag.py).  Loading torch's copy first -- without importing torch -- makes the dynamic
    loader resolve libpt_hip.so's libamdhip64 dependency to it, whatever the import order.  No torch installed (or no
    bundled runtime): the system library is used, as it is by the C++ host."""
    if "torch" in sys.modules:
        return                                    # already in the process: the loader reuses it by SONAME
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for base in (spec.submodule_search_locations or []) if spec else []:
        p = os.path.join(base, "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load():
    """dlopen libpt_hip.so and declare prototypes.  Raises if the HIP library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); path_tracer_amd has no CPU fallback")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    L.pt_create.argtypes = [C.POINTER(vp), i32]
    L.pt_destroy.argtypes = [vp]
    L.pt_destroy.restype = None
    L.pt_last_error.argtypes = [vp]
    L.pt_last_error.restype = C.c_char_p
    L.pt_resize.argtypes = [vp, i32, i32]
    L.pt_set_accum_buffer.argtypes = [vp, vp, C.c_size_t]
    L.pt_set_stream.argtypes = [vp, vp]
    L.pt_upload_mesh.argtypes = [vp, vp, i32, vp, i32]
    L.pt_upload_scene.argtypes = [vp, vp, i32, vp, i32, vp, i32]
    L.pt_set_camera.argtypes = [vp, f32p, f32p]
    L.pt_set_debug_normal.argtypes = [vp, i32]
    L.pt_set_max_bounces.argtypes = [vp, i32]
    L.pt_render_pass.argtypes = [vp, i32]
    L.pt_render.argtypes = [vp, i32, i32]
    L.pt_render_static.argtypes = [vp, i32, i32]
    L.pt_render_sum.argtypes = [vp, i32, i32]
    L.pt_scale_accum.argtypes = [vp, C.c_float, i32]
    L.pt_reset.argtypes = [vp]
    L.pt_synchronize.argtypes = [vp]
    L.pt_read_accum.argtypes = [vp, vp, C.c_size_t]
    L.pt_accum_device_ptr.argtypes = [vp]
    L.pt_accum_device_ptr.restype = vp
    L.pt_read_primary_hits.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pt_get_counters.argtypes = [vp, C.POINTER(PtCounters)]
    L.pt_reset_counters.argtypes = [vp]
    L.pt_last_render_ms.argtypes = [vp, f32p, C.POINTER(C.c_int)]
    L.pt_set_tuning.argtypes = [vp, i32, C.c_size_t, i32, i32]
    L.pt_set_option.argtypes = [vp, C.c_char_p, i32]
    L.pt_shading_preset.argtypes = [C.POINTER(PtShadingParams), i32]
    L.pt_set_shading_params.argtypes = [vp, C.POINTER(PtShadingParams)]
    L.pt_get_shading_params.argtypes = [vp, C.POINTER(PtShadingParams)]
    L.pt_bvh_selftest.argtypes = [vp, i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), f32p]
    L.pt_classify_mesh.argtypes = [vp, i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), f32p]
    L.pt_group_unique_id.argtypes = [C.c_char_p]
    L.pt_group_init.argtypes = [vp, C.c_char_p, i32, i32]
    L.pt_group_render.argtypes = [vp, i32, i32]
    L.pt_group_finalize.argtypes = [vp]
    L.pt_kernel_name.restype = C.c_char_p
    L.pt_last_kernel_name.argtypes = [vp]
    L.pt_last_kernel_name.restype = C.c_char_p
    L.pt_version.restype = i32
    _lib = L
    return L


class Renderer:
    """Headless counterpart of the reference's `Renderer` namespace + `ComputeProgram` uniform setters
    (src/Renderer.h:30-45, src/Shader.h:20-37) over the C ABI."""

    def __init__(self, width, height, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.pt_create(C.byref(h), device)
        if rc:
            raise PtError(rc, self.L.pt_last_error(None).decode())
        self.h = h
        self.screenWidth, self.screenHeight = width, height
        self._ck(self.L.pt_resize(h, width, height))

    def _ck(self, rc):
        if rc:
            raise PtError(rc, self.L.pt_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_destroy(self.h)
            self.h = None

    __del__ = close

    # --- Object: Mesh ctor + Scene::SetupSSBOs (src/Object.cpp:3-41,89-111)
    def upload_scene(self, scene):
        mt, nd = np.ascontiguousarray(scene.mesh_tris), np.ascontiguousarray(scene.nodes)
        ma, sp, tr = (np.ascontiguousarray(a) for a in (scene.materials, scene.spheres, scene.triangles))
        self._ck(self.L.pt_upload_mesh(self.h, mt.ctypes.data, len(mt), nd.ctypes.data, len(nd)))
        self._ck(self.L.pt_upload_scene(self.h, ma.ctypes.data, len(ma), sp.ctypes.data, len(sp), tr.ctypes.data, len(tr)))

    # --- Shader: SetUniformCamera / SetUniform1i (src/Shader.cpp:72-76, src/Main.cpp:128-146)
    def SetUniformCamera(self, cam):
        pos = np.ascontiguousarray(cam.position, np.float32)
        iv = np.ascontiguousarray(cam.inv_view_flat(), np.float32)
        self._ck(self.L.pt_set_camera(self.h, pos.ctypes.data_as(C.POINTER(C.c_float)), iv.ctypes.data_as(C.POINTER(C.c_float))))

    def SetUniform1i(self, name, value):
        if name == "debugNormal":
            self._ck(self.L.pt_set_debug_normal(self.h, int(value)))
        elif name == "maxBounces":
            self._ck(self.L.pt_set_max_bounces(self.h, int(value)))
        else:
            raise KeyError(name)

    # --- the dispatch (src/Main.cpp:144-150)
    def render_pass(self, p):
        self._ck(self.L.pt_render_pass(self.h, p))

    def render(self, first_pass, n_passes):
        self._ck(self.L.pt_render(self.h, first_pass, n_passes))

    def render_static(self, first_pass, n_passes):
        self._ck(self.L.pt_render_static(self.h, first_pass, n_passes))

    def render_sum(self, first_pass, n_passes):
        self._ck(self.L.pt_render_sum(self.h, first_pass, n_passes))

    def scale_accum(self, scale, set_alpha_one=True):
        self._ck(self.L.pt_scale_accum(self.h, scale, int(set_alpha_one)))

    def reset(self):
        self._ck(self.L.pt_reset(self.h))

    def synchronize(self):
        self._ck(self.L.pt_synchronize(self.h))

    def set_accum_buffer(self, ptr, nbytes):
        self._ck(self.L.pt_set_accum_buffer(self.h, ptr, nbytes))

    def set_stream(self, stream_ptr):
        self._ck(self.L.pt_set_stream(self.h, stream_ptr))

    def read_accum(self):
        out = np.empty((self.screenHeight, self.screenWidth, 4), np.float32)
        self._ck(self.L.pt_read_accum(self.h, out.ctypes.data, out.nbytes))
        return out

    def read_primary_hits(self, p):
        H, W = self.screenHeight, self.screenWidth
        ids, t = np.empty((H, W), np.int32), np.empty((H, W), np.float32)
        dirs, rng = np.empty((H, W, 3), np.float32), np.empty((H, W), np.uint32)
        self._ck(self.L.pt_read_primary_hits(self.h, p, ids.ctypes.data, t.ctypes.data, dirs.ctypes.data, rng.ctypes.data))
        return ids, t, dirs, rng

    def counters(self):
        c = PtCounters()
        self._ck(self.L.pt_get_counters(self.h, C.byref(c)))
        d = {name: int(getattr(c, name)) for name, _ in PtCounters._fields_ if name != "phase_cycles"}
        d["phase_cycles"] = [int(v) for v in c.phase_cycles]
        return d

    def reset_counters(self):
        self._ck(self.L.pt_reset_counters(self.h))

    def last_render_ms(self):
        ms = C.c_float()
        self._ck(self.L.pt_last_render_ms(self.h, C.byref(ms), None))
        return float(ms.value)

    def last_render_launches(self):
        ms, n = C.c_float(), C.c_int()
        self._ck(self.L.pt_last_render_ms(self.h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    # --- multi-GPU sample split over RCCL without torch (pt_group_*)
    @staticmethod
    def group_unique_id():
        L = load()
        buf = C.create_string_buffer(128)
        rc = L.pt_group_unique_id(buf)
        if rc:
            raise PtError(rc, L.pt_last_error(None).decode())
        return buf.raw

    def group_init(self, uid, rank, world):
        self._ck(self.L.pt_group_init(self.h, C.create_string_buffer(uid, 128), rank, world))

    def group_render(self, first_pass, total_spp):
        self._ck(self.L.pt_group_render(self.h, first_pass, total_spp))

    def group_finalize(self):
        self._ck(self.L.pt_group_finalize(self.h))

    def set_tuning(self, passes_per_item=0, workspace_limit_bytes=0, waves_per_cu=0, variant=0):
        self._ck(self.L.pt_set_tuning(self.h, passes_per_item, workspace_limit_bytes, waves_per_cu, variant))

    def kernel_name(self):
        """symbol of the trace kernel the most recent render launched (the library default before the first)"""
        return self.L.pt_last_kernel_name(self.h).decode() or self.L.pt_kernel_name().decode()

    def set_option(self, name, value):
        self._ck(self.L.pt_set_option(self.h, name.encode(), int(value)))

    def set_shading_params(self, preset=None, **fields):
        """pt_set_shading_params: start from `preset` (RT_COMP / RT_FRAG) or from the current values, override `fields`."""
        p = shading_preset(preset) if preset is not None else self.shading_params()
        for k, v in fields.items():
            if k == "sky_color":
                p.sky_color = (C.c_float * 3)(*v)
            elif k in dict(PtShadingParams._fields_):
                setattr(p, k, v)
            else:
                raise KeyError(k)
        self._ck(self.L.pt_set_shading_params(self.h, C.byref(p)))

    def shading_params(self) -> PtShadingParams:
        p = PtShadingParams()
        self._ck(self.L.pt_get_shading_params(self.h, C.byref(p)))
        return p


def classify_mesh(tris):
    """Host-only: how pt_upload_mesh sorts the triangles (pt_classify_mesh); returns (rc, in_tree, listed, never, max_growth)."""
    L = load()
    t = np.ascontiguousarray(tris)
    a, b, c, f = C.c_int(), C.c_int(), C.c_int(), C.c_float()
    rc = L.pt_classify_mesh(t.ctypes.data, len(t), C.byref(a), C.byref(b), C.byref(c), C.byref(f))
    return rc, a.value, b.value, c.value, f.value


def bvh_selftest(tris):
    """Host-only: build + verify the library's BVH over a TRIANGLE_DTYPE array; returns (rc, nodes, depth, max_leaf, sah)."""
    L = load()
    t = np.ascontiguousarray(tris)
    a, b, c, f = C.c_int(), C.c_int(), C.c_int(), C.c_float()
    rc = L.pt_bvh_selftest(t.ctypes.data, len(t), C.byref(a), C.byref(b), C.byref(c), C.byref(f))
    return rc, a.value, b.value, c.value, f.value
