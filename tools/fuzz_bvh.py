#!/usr/bin/env python3
"""Randomised exactness check of the traversal kernels on the GPU box: for each seeded random scene the image of the
brute-force kernel (variant 1: every mesh triangle behind the root-AABB gate and every sphere in index order, like
rt.comp:204-232 -- exact by construction and itself pinned against the reference's fixtures by tests/test_gpu_parity.py)
must equal, bit for bit, the images of the traversal kernels (variant 5: trees read in place, quantised nodes + 64-byte
leaf records; variant 6: trees in LDS, both workgroup sizes).  The generators aim at what a conservative tree can get
wrong: boxes flat in one axis, duplicate and degenerate triangles (ties in t, zero area), extents from 1e-4 to 1e4 in one
mesh, meshes far from the origin (quantisation step vs. float spacing), concentric and identical spheres, cameras inside
primitives.   python tools/fuzz_bvh.py [--seeds N] [--first S]      exit status 1 on any difference."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd import capi, scene as S  # noqa: E402

f32 = np.float32


def random_materials(rng, n):
    mats = []
    for _ in range(n):
        kind = rng.integers(0, 5)
        col = tuple(rng.uniform(0.05, 1.0, 3)) + (1.0,)
        if kind == 0:
            mats.append(S.material(col, emissionColor=(0, 0, 0, 0), smoothness=rng.uniform(0, 1)))
        elif kind == 1:
            mats.append(S.material(col, emissionColor=(0, 0, 0, 0), smoothness=rng.uniform(0, 1), specularSmoothness=rng.uniform(0.5, 1),
                                   specularChance=rng.uniform(0, 1)))
        elif kind == 2:
            mats.append(S.material(col, emissionColor=(0, 0, 0, 0), smoothness=rng.uniform(0.8, 1), specularSmoothness=0.99,
                                   specularChance=rng.uniform(0, 0.5), refractionAmount=rng.uniform(0.3, 1.0), ior=rng.uniform(1.0, 2.0)))
        elif kind == 3:
            mats.append(S.material(col, emissionColor=tuple(rng.uniform(0, 1, 3)) + (1.0,), emissionStrength=rng.uniform(0.5, 8)))
        else:
            mats.append(S.material(col, emissionColor=(0, 0, 0, 0)))
    return np.array(mats, dtype=S.MATERIAL_DTYPE)


def soup(rng, n, centre, spread, lo, hi):
    c = rng.uniform(-spread, spread, (n, 1, 3)) + np.asarray(centre)
    size = np.exp(rng.uniform(np.log(lo), np.log(hi), (n, 1, 1)))
    return (c + rng.normal(0, 1, (n, 3, 3)) * size).astype(f32)


def random_mesh(rng, n_mats, only_kind=None):
    kind = int(rng.integers(0, 11)) if only_kind is None else only_kind
    frame = None                                                             # generator 11: the sheets' orientation, for the camera
    n = int(rng.choice([1, 2, 3, 7, 64, 300, 1500, 6000]))
    if kind == 0:
        p = soup(rng, n, (0, 0.5, 0), 1.0, 0.02, 0.3)
    elif kind == 1:
        p = soup(rng, n, (0, 0.5, 0), 2.0, 1e-4, 10.0)                       # five decades of sizes in one mesh
    elif kind == 2:
        g = max(1, int(np.sqrt(n / 2)))                                      # axis-aligned flat grid: box of zero height
        axis = int(rng.integers(0, 3))
        u, v = np.meshgrid(np.linspace(-1, 1, g + 1), np.linspace(-1, 1, g + 1), indexing="ij")
        quad = []
        for i in range(g):
            for j in range(g):
                a, b, c, d = (u[i, j], v[i, j]), (u[i + 1, j], v[i + 1, j]), (u[i + 1, j + 1], v[i + 1, j + 1]), (u[i, j + 1], v[i, j + 1])
                quad += [[a, b, c], [a, c, d]]
        q = np.asarray(quad, f32)
        p = np.zeros((len(q), 3, 3), f32)
        others = [k for k in range(3) if k != axis]
        p[..., others[0]], p[..., others[1]] = q[..., 0], q[..., 1]
        p[..., axis] = f32(rng.uniform(0.1, 0.6))
    elif kind == 3:
        base = soup(rng, max(1, n // 3), (0, 0.5, 0), 1.0, 0.05, 0.4)        # every triangle two or three times: ties in t
        p = np.concatenate([base, base, base[: len(base) // 2]])
        p = p[rng.permutation(len(p))]
    elif kind == 4:
        off = rng.choice([100.0, 1000.0, 30000.0]) * rng.choice([-1.0, 1.0], 3)
        p = soup(rng, n, off, 1.0, 0.02, 0.3)                                # far from the origin
    elif kind == 5:
        p = np.concatenate([soup(rng, 3, (0, 0, 0), 10.0, 1e3, 1e4), soup(rng, n, (0, 0.5, 0), 1.0, 1e-3, 0.05)])
    elif kind == 6:
        p = soup(rng, n, (0, 0.5, 0), 1.0, 0.02, 0.3)
        k = rng.integers(0, len(p), max(1, len(p) // 5))
        p[k, 2] = p[k, 1]                                                    # zero-area triangles
        k = rng.integers(0, len(p), max(1, len(p) // 7))
        p[k, 1] = p[k, 0]; p[k, 2] = p[k, 0]                                 # points
    elif kind == 7:
        t = S.heightfield_mesh(int(rng.integers(2, 60)), 0).tris.copy()
        t["materialIndex"] = int(rng.integers(0, n_mats))
        return S.Mesh(t, S.Mesh.gen_bounding_box(t), int(t["materialIndex"][0]))
    elif kind == 8:
        p = soup(rng, n, (0, 0.5, 0), 0.3, 0.2, 1.5)                         # large overlapping triangles: deep, fat leaves
    elif kind == 11:
        # (only with --kind 11) the case the margins do not cover by construction: rays within 1e-3 rad of a
        # tessellated plane -- the float t of an edge-on hit is off by per cents -- with a second sheet close behind it
        g = int(rng.choice([4, 12, 30]))
        u, v = np.meshgrid(np.linspace(-1, 1, g + 1), np.linspace(-1, 1, g + 1), indexing="ij")
        quad = []
        for i in range(g):
            for j in range(g):
                a, b, c, d = (u[i, j], v[i, j]), (u[i + 1, j], v[i + 1, j]), (u[i + 1, j + 1], v[i + 1, j + 1]), (u[i, j + 1], v[i, j + 1])
                quad += [[a, b, c], [a, c, d]]
        q = np.asarray(quad, np.float64)
        flat = np.zeros((len(q), 3, 3))
        flat[..., 0], flat[..., 2] = q[..., 0], q[..., 1]
        gap = float(np.exp(rng.uniform(np.log(1e-5), np.log(1e-2))))
        both = np.concatenate([flat, flat + np.array([0.0, -gap, 0.0])])
        # a random rotation, so that the planes are not axis-aligned (boxes with volume), about the origin
        m, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        p = (both @ m.T).astype(f32)
        frame = m
    else:
        # slivers: the angle at p1 log-uniform from 1e-7 rad up (kind 9: every triangle, kind 10: one in ten), the float
        # evaluation of rt.comp:153-173 then reports hits well outside the triangle
        p = soup(rng, n, (0, 0.5, 0), 1.0, 0.05, 0.5).astype(np.float64)
        pick = np.nonzero(rng.random(len(p)) < (1.0 if kind == 9 else 0.1))[0]
        e1 = p[pick, 1] - p[pick, 0]
        perp = np.cross(e1, rng.normal(0, 1, (len(pick), 3)))
        perp /= np.linalg.norm(perp, axis=1, keepdims=True)
        ang = np.exp(rng.uniform(np.log(1e-7), np.log(0.1), (len(pick), 1)))
        p[pick, 2] = p[pick, 0] + (e1 + perp * np.linalg.norm(e1, axis=1, keepdims=True) * ang) * rng.uniform(0.3, 1.5, (len(pick), 1))
        p = p.astype(f32)
    tris = S.triangles_from_points(p, 0)
    tris["materialIndex"] = rng.integers(0, n_mats, len(tris))
    mesh = S.Mesh(tris, S.Mesh.gen_bounding_box(tris), int(tris["materialIndex"][0]))
    mesh.frame = frame
    return mesh


def random_scene(seed, only_kind=None, far_spheres=None, many_scene_tris=False, scale=1.0):
    rng = np.random.default_rng(seed)
    n_mats = int(rng.integers(1, 7))
    sc = S.Scene(materials=random_materials(rng, n_mats))
    if rng.random() < 0.9 or only_kind == 11:
        sc.mesh = random_mesh(rng, n_mats, only_kind)
    centre = np.zeros(3)
    if sc.mesh is not None and len(sc.mesh.tris):
        lo, hi = sc.mesh.nodes["boundsMin"][0][:3], sc.mesh.nodes["boundsMax"][0][:3]
        centre = np.clip((lo + hi) / 2, -5e4, 5e4)
        if np.max(hi - lo) > 100:                                            # the huge-triangle meshes: look at the small ones
            centre = np.array([0, 0.5, 0.0])
    n_sph = int(rng.choice([0, 0, 1, 2, 5, 40, 250]))
    # small spheres seen from far away: rt.comp:126-146 then reports hits outside the sphere (hb^2 - a c is rounding noise)
    far = (rng.random() < 0.15) if far_spheres is None else far_spheres
    if far and n_sph == 0:
        n_sph = 40
    for i in range(n_sph):
        mode = rng.integers(0, 5)
        if mode == 0 and len(sc.spheres):
            j = int(rng.integers(0, len(sc.spheres)))                        # concentric with / identical to an earlier one
            pos = sc.spheres["position"][j]
            rad = float(sc.spheres["radius"][j]) * float(rng.choice([1.0, 0.5, 2.0]))
        else:
            pos = centre + rng.uniform(-1.5, 1.5, 3)
            rad = float(np.exp(rng.uniform(np.log(1e-4), np.log(1e-2)))) if far else float(np.exp(rng.uniform(np.log(0.01), np.log(1.0))))
        sc.add_sphere(pos, rad, int(rng.integers(0, n_mats)))
    n_tri = int(rng.choice([0, 1, 2, 4]))
    for i in range(n_tri):
        if i == 0 and rng.random() < 0.7:
            sc.add_triangle(centre + (5000.0, -0.3, 5000.0), centre + (-5000.0, -0.3, 5000.0), centre + (0.0, -0.3, -5000.0), int(rng.integers(0, n_mats)))
        else:
            c = centre + rng.uniform(-2, 2, 3)
            sc.add_triangle(c + rng.normal(0, 1, 3), c + rng.normal(0, 1, 3), c + rng.normal(0, 1, 3), int(rng.integers(0, n_mats)))
    # camera: around the centre looking roughly at it, sometimes in the middle of everything
    dist = float(rng.choice([4.0, 20.0, 100.0])) if far else float(rng.choice([0.0, 0.5, 2.0, 4.0]))
    yaw, pitch = float(rng.uniform(0, 360)), float(rng.uniform(-60, 60))
    d = np.array([np.cos(np.radians(yaw)) * np.cos(np.radians(pitch)), np.sin(np.radians(pitch)), np.sin(np.radians(yaw)) * np.cos(np.radians(pitch))])
    cam = S.Camera(position=(centre - d * dist).astype(f32), yaw=yaw, pitch=pitch).UpdateView()
    if only_kind == 11:
        # the camera a hair above the upper sheet, outside it, looking along it: primary rays at 1e-6 .. 1e-2 rad to the plane
        m = sc.mesh.frame
        along = m @ np.array([np.cos(0.3), 0.0, np.sin(0.3)])
        up = m @ np.array([0.0, 1.0, 0.0])
        h = float(np.exp(rng.uniform(np.log(1e-6), np.log(1e-2))))
        pos = -along * 1.6 + up * h
        yaw = float(np.degrees(np.arctan2(along[2], along[0]))); pitch = float(np.degrees(np.arcsin(np.clip(along[1], -1, 1))))
        cam = S.Camera(position=pos.astype(f32), yaw=yaw, pitch=pitch).UpdateView()
    bounces = int(rng.choice([1, 2, 4, 8, 12]))
    if many_scene_tris:
        # more scene triangles than the kernels stage in LDS (pt_kernels.hpp kSceneLdsTris = 16): the scalar-load route.
        # A random stream of its own, so that the seeds above keep naming the same scenes.
        r2 = np.random.default_rng(seed + 2 * 10**9)
        for _ in range(int(r2.integers(17, 40))):
            c = centre + r2.uniform(-2, 2, 3)
            sc.add_triangle(c + r2.normal(0, 0.5, 3), c + r2.normal(0, 0.5, 3), c + r2.normal(0, 0.5, 3), int(r2.integers(0, n_mats)))
    if scale != 1.0:
        # the whole scene and the camera position times a power of ten: products near overflow (1e16) or in the denormals (1e-12)
        k = f32(scale)
        if sc.mesh is not None:
            t = sc.mesh.tris.copy()
            for name in ("p1", "p2", "p3"):
                t[name][:, :3] *= k
            sc.mesh = S.Mesh(t, S.Mesh.gen_bounding_box(t), sc.mesh.materialIndex)
        for name in ("p1", "p2", "p3"):
            sc.triangles[name][:, :3] *= k
        sc.spheres["position"] *= k
        sc.spheres["radius"] *= k
        # the view rotation as it is (glm::lookAt of a position of 1e12 is 0/0, in the reference too); only the position moves
        cam.position = (np.asarray(cam.position, np.float64) * scale).astype(f32)
    return sc, cam, bounces


def render(sc, cam, W, H, bounces, spp, variant, lds_block=0, no_fused=False, preset=0):
    r = capi.Renderer(W, H)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    r.set_tuning(0, 0, 0, variant)
    if preset:
        r.set_shading_params(preset=preset)           # 1: the rt.frag literal set (hit limit 0, two draws, ...): the GEN kernels
    if lds_block:
        r.set_option("lds_block", lds_block)
    if no_fused:
        r.set_option("no_fused", 1)                   # the per-pass workspace + pt_accumulate route (what long renders take)
    r.render(1, spp)
    img, name = r.read_accum(), r.kernel_name()
    r.close()
    return img, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--size", type=int, nargs=2, default=[128, 72])
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--kind", type=int, default=None, help="mesh generator (0..10), default: random per seed")
    ap.add_argument("--preset", type=int, default=0, help="shading preset (0 rt.comp, 1 rt.frag)")
    ap.add_argument("--many-scene-tris", action="store_true", help="17..39 more scene triangles per scene (beyond the LDS-staged 16)")
    ap.add_argument("--scale", type=float, default=1.0, help="multiply every coordinate, radius and the camera position by this")
    ap.add_argument("--far-spheres", action="store_true", help="every scene: small spheres (r 1e-4..1e-2) seen from 4..100 away")
    a = ap.parse_args()
    capi.load()
    W, H = a.size
    bad, kernels, finite = 0, {}, 0
    for seed in range(a.first, a.first + a.seeds):
        sc, cam, bounces = random_scene(seed, a.kind, True if a.far_spheres else None, a.many_scene_tris, a.scale)
        want, _ = render(sc, cam, W, H, bounces, a.spp, 1, 0, False, a.preset)
        finite += int(np.isfinite(want).all())
        for variant, blk in ((5, 0), (6, 512), (6, 1024)):
            got, name = render(sc, cam, W, H, bounces, a.spp, variant, blk, a.no_fused, a.preset)
            kernels[name] = kernels.get(name, 0) + 1
            n = int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
            if n:
                bad += 1
                print(f"seed {seed}: variant {variant} lds_block {blk} ({name}) differs from brute force in {n} pixels; "
                      f"mesh {len(sc.mesh_tris)} tris, {len(sc.spheres)} spheres, {len(sc.triangles)} scene tris, {bounces} bounces", flush=True)
        if (seed - a.first) % 25 == 24:
            print(f"... {seed - a.first + 1} scenes, {bad} differences so far", flush=True)
    print(f"fuzz: {a.seeds} scenes from seed {a.first}, {a.size[0]}x{a.size[1]}, {a.spp} spp: {bad} differing renders; "
          f"{finite} brute-force images all finite; kernels: {kernels}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
