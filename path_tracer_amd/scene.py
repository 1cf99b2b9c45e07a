"""Host-side scene model: Python mirror of the reference's `Object`/`Camera` surface.

Mirrors (names, field meaning, defaults) of:
  * Material / Sphere / Triangle / Node / Scene / Mesh  -- src/Object.h:8-93, src/Object.cpp:43-181
  * Camera.UpdateView                                   -- src/Camera.cpp:13-26 (GLM 0.9.8.5 lookAtRH + inverse)
  * the hard-coded default scene of main()              -- src/Main.cpp:20-74
The records are numpy structured dtypes that are byte-identical to the std430 / C++ layouts, so the
arrays can be handed straight to the C ABI (`include/pt_hip.h`).  This is the thin Python twin of
the C++ host classes in `path_tracer_amd/host/`; it holds no device code and no CPU renderer.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

f32 = np.float32

# --- std430 records (src/Object.h:8-75, rt.comp:33-48) ------------------------------------------
MATERIAL_DTYPE = np.dtype([
    ("baseColor", "<f4", 4), ("specularColor", "<f4", 4), ("emissionColor", "<f4", 4),
    ("smoothness", "<f4"), ("specularSmoothness", "<f4"), ("emissionStrength", "<f4"),
    ("ior", "<f4"), ("refractionAmount", "<f4"), ("specularChance", "<f4"), ("pad", "<i4", 2)])
SPHERE_DTYPE = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("materialIndex", "<u4"), ("pad", "<i4", 3)])
TRIANGLE_DTYPE = np.dtype([("p1", "<f4", 4), ("p2", "<f4", 4), ("p3", "<f4", 4), ("materialIndex", "<u4"), ("pad", "<i4", 3)])
NODE_DTYPE = np.dtype([("boundsMin", "<f4", 4), ("boundsMax", "<f4", 4),
                       ("triIndex", "<i4"), ("numTris", "<i4"), ("childrenIndex", "<i4"), ("pad", "<i4")])
assert MATERIAL_DTYPE.itemsize == 80 and SPHERE_DTYPE.itemsize == 32
assert TRIANGLE_DTYPE.itemsize == 64 and NODE_DTYPE.itemsize == 48


def material(baseColor=(1, 1, 1, 1), specularColor=(1, 1, 1, 1), emissionColor=(1, 1, 1, 1), smoothness=0.0,
             specularSmoothness=0.0, emissionStrength=0.0, ior=1.5, refractionAmount=0.0, specularChance=0.0):
    """One `Material` with the C++ default member initialisers (src/Object.h:10-18)."""
    m = np.zeros((), MATERIAL_DTYPE)
    m["baseColor"], m["specularColor"], m["emissionColor"] = baseColor, specularColor, emissionColor
    m["smoothness"], m["specularSmoothness"], m["emissionStrength"] = smoothness, specularSmoothness, emissionStrength
    m["ior"], m["refractionAmount"], m["specularChance"] = ior, refractionAmount, specularChance
    return m


def default_materials() -> np.ndarray:
    """The five materials pushed in main() (src/Main.cpp:20-69): specular, diffuse, glass, ground, sun."""
    z4 = (0, 0, 0, 0)
    return np.array([
        material((0.2, 0.9, 0.1, 1.0), emissionColor=z4, smoothness=0.9),
        material((0.9, 0.1, 0.1, 1.0), emissionColor=z4, smoothness=0.2, specularSmoothness=0.99, specularChance=0.1),
        material((0.9, 0.1, 0.1, 1.0), emissionColor=z4, smoothness=0.99, specularSmoothness=0.99, specularChance=0.2,
                 refractionAmount=0.8, ior=1.05),
        material((0.9, 0.9, 0.9, 1.0), emissionColor=z4),
        material((1.0, 1.0, 1.0, 1.0), emissionColor=(1, 1, 1, 1), emissionStrength=5.0),
    ], dtype=MATERIAL_DTYPE)


def triangles_from_points(p: np.ndarray, material_index) -> np.ndarray:
    """(N,3,3) float32 points -> Triangle[] with w = 0 (src/Object.cpp:59-67,152-160)."""
    p = np.asarray(p, f32).reshape(-1, 3, 3)
    t = np.zeros(len(p), TRIANGLE_DTYPE)
    t["p1"][:, :3], t["p2"][:, :3], t["p3"][:, :3] = p[:, 0], p[:, 1], p[:, 2]
    t["materialIndex"] = material_index
    return t


def load_obj(path: str):
    """`Mesh::Load` (src/Object.cpp:113-165): only `v x y z` and `f a b c` records, 1-based indices.
    Also accepts `f a/b/c` tokens.  Returns (vertices (V,3) f32, faces (F,3) i32)."""
    verts, faces = [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                x, y, z = line.split()[1:4]
                verts.append((f32(x), f32(y), f32(z)))
            elif line.startswith("f "):
                a, b, c = (int(tok.split("/")[0]) - 1 for tok in line.split()[1:4])
                faces.append((a, b, c))
    return np.asarray(verts, f32).reshape(-1, 3), np.asarray(faces, np.int32).reshape(-1, 3)


@dataclass
class Mesh:
    """`Mesh` (src/Object.h:77-93): triangle soup + root AABB node (`GenBoundingBox`, src/Object.cpp:167-181)."""
    tris: np.ndarray
    nodes: np.ndarray
    materialIndex: int = 0

    @staticmethod
    def from_arrays(vertices: np.ndarray, faces: np.ndarray, materialIndex: int) -> "Mesh":
        pts = np.asarray(vertices, f32)[np.asarray(faces, np.int64)]
        tris = triangles_from_points(pts, materialIndex)
        return Mesh(tris, Mesh.gen_bounding_box(tris), materialIndex)

    @staticmethod
    def from_obj(path: str, materialIndex: int) -> "Mesh":
        v, fidx = load_obj(path)
        return Mesh.from_arrays(v, fidx, materialIndex)

    @staticmethod
    def merged(parts) -> "Mesh":
        """Several meshes as ONE mesh for SSBO 1/2: in the reference every `Mesh` constructor rebinds
        the two buffers, so only the mesh built last is traced (src/Object.cpp:104-110).  Here the triangles are
        concatenated in the order given, each keeping its own materialIndex, behind one root AABB over all of them --
        for the shader that is simply a bigger mesh (rt.comp:213-218), so the oracle and the kernels need nothing new."""
        parts = list(parts)
        tris = np.concatenate([m.tris for m in parts]) if parts else np.zeros(0, TRIANGLE_DTYPE)
        return Mesh(tris, Mesh.gen_bounding_box(tris), parts[0].materialIndex if parts else 0)

    @staticmethod
    def gen_bounding_box(tris: np.ndarray) -> np.ndarray:
        node = np.zeros(1, NODE_DTYPE)
        if len(tris):
            allp = np.concatenate([tris["p1"], tris["p2"], tris["p3"]])      # includes w = 0 (src/Object.cpp:77-87)
            node["boundsMin"][0] = np.minimum(allp.min(0), f32(1e30))
            node["boundsMax"][0] = np.maximum(allp.max(0), f32(-1e30))
        else:
            node["boundsMin"][0], node["boundsMax"][0] = f32(1e30), f32(-1e30)
        node["numTris"], node["childrenIndex"] = len(tris), 1
        return node


@dataclass
class Scene:
    """`Scene` (src/Object.h:50-61) plus the (last constructed) mesh bound to SSBO 1/2."""
    materials: np.ndarray = field(default_factory=lambda: np.zeros(0, MATERIAL_DTYPE))
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, SPHERE_DTYPE))
    triangles: np.ndarray = field(default_factory=lambda: np.zeros(0, TRIANGLE_DTYPE))
    mesh: Mesh | None = None

    def add_sphere(self, pos, radius, materialIndex):
        s = np.zeros(1, SPHERE_DTYPE)
        s["position"][0], s["radius"], s["materialIndex"] = np.asarray(pos, f32), f32(radius), materialIndex
        self.spheres = np.concatenate([self.spheres, s])

    def add_triangle(self, p1, p2, p3, materialIndex):
        self.triangles = np.concatenate([self.triangles, triangles_from_points([[p1, p2, p3]], materialIndex)])

    @property
    def mesh_tris(self):
        return self.mesh.tris if self.mesh is not None else np.zeros(0, TRIANGLE_DTYPE)

    @property
    def nodes(self):
        return self.mesh.nodes if self.mesh is not None else Mesh.gen_bounding_box(np.zeros(0, TRIANGLE_DTYPE))

    def num_primitives(self):
        return len(self.mesh_tris) + len(self.triangles) + len(self.spheres)


# --- camera (src/Camera.h:8-29, src/Camera.cpp:13-26) ------------------------------------------
def _dot(a, b):
    # GLM 0.9.8.5 compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z
    t = a * b
    return f32(f32(t[0] + t[1]) + t[2])


def _cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], f32)


def _normalize(v):
    return (v * (f32(1) / np.sqrt(_dot(v, v)))).astype(f32)


def _glm_inverse(m: np.ndarray) -> np.ndarray:
    """glm::inverse(mat4) 0.9.8.5 (detail/func_matrix.inl compute_inverse<tmat4x4>), float32 op order.
    m[c][r] column-major."""
    c = lambda a, b, c_, d: f32(f32(a * b) - f32(c_ * d))
    C00 = c(m[2][2], m[3][3], m[3][2], m[2][3]); C02 = c(m[1][2], m[3][3], m[3][2], m[1][3]); C03 = c(m[1][2], m[2][3], m[2][2], m[1][3])
    C04 = c(m[2][1], m[3][3], m[3][1], m[2][3]); C06 = c(m[1][1], m[3][3], m[3][1], m[1][3]); C07 = c(m[1][1], m[2][3], m[2][1], m[1][3])
    C08 = c(m[2][1], m[3][2], m[3][1], m[2][2]); C10 = c(m[1][1], m[3][2], m[3][1], m[1][2]); C11 = c(m[1][1], m[2][2], m[2][1], m[1][2])
    C12 = c(m[2][0], m[3][3], m[3][0], m[2][3]); C14 = c(m[1][0], m[3][3], m[3][0], m[1][3]); C15 = c(m[1][0], m[2][3], m[2][0], m[1][3])
    C16 = c(m[2][0], m[3][2], m[3][0], m[2][2]); C18 = c(m[1][0], m[3][2], m[3][0], m[1][2]); C19 = c(m[1][0], m[2][2], m[2][0], m[1][2])
    C20 = c(m[2][0], m[3][1], m[3][0], m[2][1]); C22 = c(m[1][0], m[3][1], m[3][0], m[1][1]); C23 = c(m[1][0], m[2][1], m[2][0], m[1][1])
    F0 = np.array([C00, C00, C02, C03], f32); F1 = np.array([C04, C04, C06, C07], f32); F2 = np.array([C08, C08, C10, C11], f32)
    F3 = np.array([C12, C12, C14, C15], f32); F4 = np.array([C16, C16, C18, C19], f32); F5 = np.array([C20, C20, C22, C23], f32)
    V0 = np.array([m[1][0], m[0][0], m[0][0], m[0][0]], f32); V1 = np.array([m[1][1], m[0][1], m[0][1], m[0][1]], f32)
    V2 = np.array([m[1][2], m[0][2], m[0][2], m[0][2]], f32); V3 = np.array([m[1][3], m[0][3], m[0][3], m[0][3]], f32)
    I0 = (V1 * F0 - V2 * F1) + V3 * F2
    I1 = (V0 * F0 - V2 * F3) + V3 * F4
    I2 = (V0 * F1 - V1 * F3) + V3 * F5
    I3 = (V0 * F2 - V1 * F4) + V2 * F5
    sa = np.array([+1, -1, +1, -1], f32); sb = np.array([-1, +1, -1, +1], f32)
    inv = np.stack([I0 * sa, I1 * sb, I2 * sa, I3 * sb]).astype(f32)          # inv[c][r]
    row0 = np.array([inv[0][0], inv[1][0], inv[2][0], inv[3][0]], f32)
    d0 = (np.asarray(m[0], f32) * row0).astype(f32)
    det = f32(f32(d0[0] + d0[1]) + f32(d0[2] + d0[3]))
    return (inv * (f32(1) / det)).astype(f32)


@dataclass
class Camera:
    """`Camera` state (src/Camera.h:10-23).  `UpdateView` follows src/Camera.cpp:13-26 op for op in
    float32 (radians/cos/sin through float32 libm, GLM lookAtRH, GLM general cofactor inverse)."""
    position: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    yaw: float = 0.0
    pitch: float = 0.0
    speed: float = 2.0
    moving: bool = False
    forward: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    up: np.ndarray = field(default_factory=lambda: np.array([0, 1, 0], f32))
    right: np.ndarray = field(default_factory=lambda: np.array([1, 0, 0], f32))
    inverseView: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=f32))

    def UpdateView(self):
        rad = lambda deg: f32(f32(deg) * f32(0.01745329251994329576923690768489))
        cy, sy = f32(np.cos(rad(self.yaw))), f32(np.sin(rad(self.yaw)))
        cp, sp = f32(np.cos(rad(self.pitch))), f32(np.sin(rad(self.pitch)))
        direction = np.array([cy * cp, sp, sy * cp], f32)
        self.position = np.asarray(self.position, f32)
        self.forward = _normalize(direction)
        self.right = _normalize(_cross(np.array([0, 1, 0], f32), self.forward))
        self.up = _cross(self.forward, self.right)
        eye, center = self.position, (self.position + self.forward).astype(f32)
        # glm::lookAtRH (gtc/matrix_transform.inl:521-546)
        fwd = _normalize((center - eye).astype(f32))
        s = _normalize(_cross(fwd, self.up))
        u = _cross(s, fwd)
        view = np.eye(4, dtype=f32)                                            # view[c][r]
        view[0][0], view[1][0], view[2][0] = s
        view[0][1], view[1][1], view[2][1] = u
        view[0][2], view[1][2], view[2][2] = -fwd
        view[3][0], view[3][1], view[3][2] = -_dot(s, eye), -_dot(u, eye), _dot(fwd, eye)
        self.inverseView = _glm_inverse(view)
        return self

    def inv_view_flat(self) -> np.ndarray:
        """16 floats, column-major, as `glUniformMatrix4fv(..., GL_FALSE, ...)` receives them (src/Shader.cpp:72-76)."""
        return np.ascontiguousarray(self.inverseView, f32).reshape(16)


# --- canned scenes (BASELINE.json configs) -----------------------------------------
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def bunny_mesh(materialIndex: int) -> Mesh:
    """bunny1 (330 verts / 656 tris), the mesh main() loads (src/Main.cpp:71).  Stored as parsed arrays
    (data/bunny1.npz, written by oracle/gen_goldens.py from the reference's res/meshes/bunny1.obj)."""
    z = np.load(os.path.join(_DATA, "bunny1.npz"))
    return Mesh.from_arrays(z["vertices"], z["faces"], materialIndex)


def _light_and_ground(sc: Scene):
    sc.add_triangle((-1.0, 0.0, 3.0), (1.0, 0.0, 3.0), (0.0, 1.4, 3.0), 4)                    # src/Main.cpp:73
    sc.add_triangle((5000.0, 0.0, 5000.0), (-5000.0, 0.0, 5000.0), (0.0, 0.0, -5000.0), 3)    # src/Main.cpp:74


def default_scene(mesh: Mesh | None = None) -> Scene:
    """The scene main() builds (src/Main.cpp:20-74): bunny1 material 3 + light + ground, no spheres."""
    sc = Scene(materials=default_materials(), mesh=mesh if mesh is not None else bunny_mesh(3))
    _light_and_ground(sc)
    return sc


def mixed_scene(mesh: Mesh | None = None) -> Scene:
    """BASELINE config 5 / SURVEY App. C.5: bunny material 1, the 12x1x12 glass-sphere grid and the
    specular sphere that main() carries commented out (src/Main.cpp:76-90), light + ground."""
    sc = Scene(materials=default_materials(), mesh=mesh if mesh is not None else bunny_mesh(1))
    _light_and_ground(sc)
    for i in range(-6, 6):
        for k in range(-6, 6):
            # ((float)i) / 3.0 is a double divide narrowed to float (src/Main.cpp:82)
            sc.add_sphere((f32(float(f32(i)) / 3.0), f32(float(f32(0)) + 0.101), f32(float(f32(k)) / 3.0)), 0.1, 2)
    sc.add_sphere((-0.2, 1.0, 0.0), 0.5, 0)                                                    # src/Main.cpp:90
    return sc


def heightfield_mesh(n: int, materialIndex: int = 3) -> Mesh:
    """Synthetic mesh for BASELINE config 3: n x n grid over [-1,1]^2,
    y = 0.1*sin(7x)*cos(5z) + 0.2, two triangles per cell (n=708 -> 1,002,528 triangles)."""
    g = np.linspace(-1.0, 1.0, n + 1, dtype=np.float64)
    x, z = np.meshgrid(g, g, indexing="xy")
    y = 0.1 * np.sin(7 * x) * np.cos(5 * z) + 0.2
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(f32)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, b, c], -1), np.stack([b, d, c], -1)]).astype(np.int32)
    return Mesh.from_arrays(verts, faces, materialIndex)


def bench_camera() -> Camera:
    """Bench pose for configs 1-4: the reference default (0,0,0) sits inside the bunny."""
    return Camera(position=np.array([1.5, 0.4, 0.0], f32), yaw=0.0, pitch=0.0).UpdateView()


def mixed_camera() -> Camera:
    return Camera(position=np.array([1.5, 0.8, 1.5], f32), yaw=45.0, pitch=20.0).UpdateView()


def heightfield_camera() -> Camera:
    """Bench pose for config 3: above the height field, looking down at it (most primary rays traverse the mesh)."""
    return Camera(position=np.array([0.2, 1.6, 2.2], f32), yaw=95.0, pitch=35.0).UpdateView()
