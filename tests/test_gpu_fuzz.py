"""Randomised and adversarial scenes: the BVH traversal kernels (variants 5/6) against the brute-force kernel, which
scans every primitive exactly like the reference shader, and against the CPU oracle.  The adversarial cases aim at the one
situation the traversal does not cover by construction: rays (nearly) coplanar with mesh triangles."""
import numpy as np
import pytest

from conftest import bits_equal, n_diff_pixels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as c
    c.load()
    return c


def render(capi, sc, cam, W, H, bounces, spp, variant):
    r = capi.Renderer(W, H)
    r.upload_scene(sc)
    r.SetUniformCamera(cam)
    r.SetUniform1i("maxBounces", bounces)
    r.set_tuning(0, 0, 0, variant)
    r.reset_counters()
    r.render(1, spp)
    img, c = r.read_accum(), r.counters()
    r.close()
    return img, c


def random_scene(seed):
    from path_tracer_amd import scene as S
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 400))
    # a triangle soup: clustered small triangles + a few large ones, some exactly duplicated, some sharing vertices
    centres = rng.uniform(-1.0, 1.0, (n, 1, 3)) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, 0.6, 0.0])
    size = np.where(rng.random((n, 1, 1)) < 0.9, 0.15, 0.8)
    pts = (centres + rng.uniform(-1.0, 1.0, (n, 3, 3)) * size).astype(np.float32)
    pts[n // 2:n // 2 + n // 8] = pts[:n // 8]                       # duplicates: lowest index must win the ties
    pts[-1, 0] = pts[0, 0]; pts[-1, 1] = pts[0, 1]                    # shared edge
    tris = S.triangles_from_points(pts, int(rng.integers(0, 5)))
    tris["materialIndex"] = rng.integers(0, 5, n).astype(np.uint32)
    mesh = S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 0)
    sc = S.Scene(materials=S.default_materials(), mesh=mesh)
    sc.add_triangle((-50, 0, 50), (50, 0, 50), (0, 0, -50), 3)
    sc.add_triangle((-1, 0, 3), (1, 0, 3), (0, 1.4, 3), 4)
    for _ in range(int(rng.integers(0, 40))):
        sc.add_sphere(rng.uniform(-1.5, 1.5, 3) * np.array([1, 0.5, 1]) + np.array([0, 0.7, 0]), float(rng.uniform(0.03, 0.4)), int(rng.integers(0, 5)))
    cam = S.Camera(position=(rng.uniform(-2.5, 2.5, 3) * np.array([1, 0.4, 1]) + np.array([0, 1.0, 0])).astype(np.float32),
                   yaw=float(rng.uniform(0, 360)), pitch=float(rng.uniform(-40, 25))).UpdateView()
    return sc, cam


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_all_variants_and_oracle(capi, seed):
    import oracle_py as O
    sc, cam = random_scene(seed)
    W, H, B, SPP = 96, 56, 6, 3
    ref, c_ref = render(capi, sc, cam, W, H, B, SPP, 1)
    assert c_ref["bvh_tri_tests"] == 0                 # variant 1 = brute force (some cameras look away from the mesh)
    for variant in (5, 6):
        img, c = render(capi, sc, cam, W, H, B, SPP, variant)
        assert c["mesh_tri_tests"] == 0 and (c["bvh_tri_tests"] > 0 or c_ref["mesh_segments"] == 0), variant
        assert bits_equal(img, ref), (seed, variant, n_diff_pixels(img, ref))
    want = O.Oracle(sc, cam, W, H, B).render(1, SPP)
    assert n_diff_pixels(ref, want) == 0, seed


def flat_grid(n, y=0.0):
    from path_tracer_amd import scene as S
    g = np.linspace(-1.0, 1.0, n + 1)
    x, z = np.meshgrid(g, g, indexing="xy")
    verts = np.stack([x, np.full_like(x, y), z], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, b, c], -1), np.stack([b, d, c], -1)])
    return S.Mesh.from_arrays(verts, faces, 1)


@pytest.mark.parametrize("cam_y", [0.0, 1e-7, 1e-5, 1e-3, 0.25])
def test_rays_coplanar_with_a_flat_mesh(capi, cam_y):
    """camera in (or a hair above) the plane of a flat 2048-triangle mesh, looking along it: every primary ray is
    (nearly) coplanar with every triangle -- the reference's u, v are noise / noise there.  The traversal must still
    return what the full scan returns (4 M primary rays + their bounces per case)."""
    from path_tracer_amd import scene as S
    flat = flat_grid(32, 0.0)
    # one small triangle off the plane: a perfectly flat mesh has a zero-thickness root AABB, which the reference's gate
    # (rt.comp:213) lets almost no ray through
    lid = S.triangles_from_points([[(-0.9, 0.4, -0.9), (-0.8, 0.4, -0.9), (-0.9, 0.4, -0.8)]], 1)
    tris = np.concatenate([flat.tris, lid])
    mesh = S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 1)
    sc = S.Scene(materials=S.default_materials(), mesh=mesh)
    sc.add_triangle((-1, 0.0, 3), (1, 0.0, 3), (0, 1.4, 3), 4)
    sc.add_sphere((0.2, 0.3, 0.1), 0.3, 0)
    cam = S.Camera(position=np.array([2.5, cam_y, 0.05], np.float32), yaw=0.0, pitch=0.0).UpdateView()   # yaw 0 looks down -x (bench pose)
    W, H, B, SPP = 512, 256, 5, 32
    ref, c_ref = render(capi, sc, cam, W, H, B, SPP, 1)
    assert c_ref["mesh_segments"] > 0.1 * c_ref["samples"]
    for variant in (5, 6):
        img, c = render(capi, sc, cam, W, H, B, SPP, variant)
        assert bits_equal(img, ref), (cam_y, variant, n_diff_pixels(img, ref))


def test_axis_parallel_rays_and_degenerate_triangles(capi):
    """rays with exactly zero direction components (0 * inf in the slab test), zero-area and sliver triangles"""
    import oracle_py as O
    from path_tracer_amd import scene as S
    rng = np.random.default_rng(5)
    pts = rng.uniform(-0.5, 0.5, (120, 3, 3)).astype(np.float32) + np.array([0, 0.5, 0], np.float32)
    pts[10, 1] = pts[10, 0]                                           # zero area: two equal vertices
    pts[11, 2] = (pts[11, 0] + pts[11, 1]) / 2                        # zero area: collinear
    pts[12] = pts[12, 0]                                              # a point
    pts[13, :, 0] = 0.25                                              # in the plane x = 0.25 (the camera's too)
    tris = S.triangles_from_points(pts, 1)
    sc = S.Scene(materials=S.default_materials(), mesh=S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 1))
    sc.add_triangle((-50, 0, 50), (50, 0, 50), (0, 0, -50), 3)
    # yaw 90: forward = (cos 90deg, 0, sin 90deg) has a float-exact zero / near-zero x; camera in the plane x = 0.25
    cam = S.Camera(position=np.array([0.25, 0.5, -2.0], np.float32), yaw=90.0, pitch=0.0).UpdateView()
    W, H, B, SPP = 128, 72, 4, 4
    ref, _ = render(capi, sc, cam, W, H, B, SPP, 1)
    for variant in (5, 6):
        img, _ = render(capi, sc, cam, W, H, B, SPP, variant)
        assert bits_equal(img, ref), (variant, n_diff_pixels(img, ref))
    assert n_diff_pixels(ref, O.Oracle(sc, cam, W, H, B).render(1, SPP)) == 0


@pytest.mark.parametrize("gap", [4e-6, 8e-6, 2e-5])
def test_sphere_hits_at_negative_distance(capi, gap):
    """HitSphere counts t1 > -HIT_LIMIT (rt.comp:142): a camera `gap` outside a sphere, looking AWAY from it, "hits" the
    sphere at t = -gap..0 when gap < 1e-5 (the closest hit of all: t < 0), and the path continues from behind the camera.
    A hit record that assumes t > 0 breaks here (it did: 46 pixels of config 5)."""
    import oracle_py as O
    from path_tracer_amd import scene as S
    rng = np.random.default_rng(11)
    pts = rng.uniform(-0.3, 0.3, (40, 3, 3)).astype(np.float32) + np.array([0, 0.4, 2.5], np.float32)
    tris = S.triangles_from_points(pts, 1)
    sc = S.Scene(materials=S.default_materials(), mesh=S.Mesh(tris, S.Mesh.gen_bounding_box(tris), 1))
    sc.add_triangle((-50, 0, 50), (50, 0, 50), (0, 0, -50), 3)
    sc.add_sphere((0.0, 1.0, 0.0), 1.0, 2)                           # glass; the camera sits just outside its +z pole
    sc.add_sphere((0.0, 1.0, 0.0), 0.999999, 0)                      # a second shell: two negative distances to order
    for k in range(6):
        sc.add_sphere((0.5 * k - 1.2, 0.3, 3.0), 0.2, k % 5)
    cam = S.Camera(position=np.array([0.0, 1.0, np.float32(1.0) + np.float32(gap)], np.float32), yaw=-90.0, pitch=0.0).UpdateView()   # looks along +z
    W, H, B, SPP = 128, 72, 6, 3
    want = O.Oracle(sc, cam, W, H, B)
    ids, t, _, _ = want.primary_hits(1)
    if gap < 1e-5:
        assert (t[ids >= 0] < 0).sum() > 1000, "the case under test: primary hits at negative distance"
    ref, _ = render(capi, sc, cam, W, H, B, SPP, 1)
    for variant in (5, 6):
        img, _ = render(capi, sc, cam, W, H, B, SPP, variant)
        assert bits_equal(img, ref), (variant, n_diff_pixels(img, ref))
    assert n_diff_pixels(ref, want.render(1, SPP)) == 0


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_exactly_zero_direction_component_reaches_the_traversal(capi, axis):
    """An inverseView whose row `axis` is all zeros makes EVERY primary ray's direction component on that axis exactly
    (+-)0 (rt.comp:333 gives 0*X + 0 + 0*Y), with the camera inside that axis' slab of the mesh and of most BVH boxes: the
    folded slab test of the traversal kernels sees 1/d = +-inf there (b*inf - o*inf = inf - inf on one plane only).  The
    traversal must still return what the full scan and the oracle return."""
    import oracle_py as O
    from path_tracer_amd import scene as S
    sc = S.default_scene()
    # a fan of rays in the plane {axis = const} through the bunny (bounds about [-0.24, 0.25] x [0, 0.49] x [-0.22, 0.17])
    pos = {0: (0.011, 0.25, 1.2), 1: (1.2, 0.2, 0.013), 2: (1.2, 0.25, 0.017)}[axis]
    cols = {0: [(0, 0, 0.9), (0, 1, 0), (0, -0.1, -1)],     # X sweeps z a little, Y sweeps y, forward -z
            1: [(0, 0, 1), (0.4, 0, 0), (-1, 0, 0.02)],     # Y sweeps x, forward -x, all in the plane y = 0.2
            2: [(0.3, 0, 0), (0, 1, 0), (-1, -0.05, 0)]}[axis]
    cam = S.Camera(position=np.array(pos, np.float32), yaw=0.0, pitch=0.0).UpdateView()
    iv = np.zeros((4, 4), np.float32)
    for k in range(3):
        iv[k, :3] = cols[k]
    iv[3] = (*pos, 1.0)
    assert np.all(iv[:3, axis] == 0.0)
    cam.inverseView = iv
    W, H, B, SPP = 128, 72, 4, 4
    ref, c_ref = render(capi, sc, cam, W, H, B, SPP, 1)
    assert c_ref["mesh_segments"] > 0.2 * c_ref["samples"]
    ids = None
    for variant in (5, 6):
        img, c = render(capi, sc, cam, W, H, B, SPP, variant)
        assert c["bvh_tri_tests"] > 0
        assert bits_equal(img, ref), (axis, variant, n_diff_pixels(img, ref))
    want = O.Oracle(sc, cam, W, H, B).render(1, SPP)
    assert n_diff_pixels(ref, want) == 0
    # the primary rays really have the exact zero, and a good share of them hit the mesh
    r = capi.Renderer(W, H)
    r.upload_scene(sc); r.SetUniformCamera(cam)
    ids, _, dirs, _ = r.read_primary_hits(1)
    r.close()
    assert np.all(dirs[..., axis] == 0.0)
    assert ((ids >= 0) & (ids < len(sc.mesh_tris))).mean() > 0.05


def _same_or_both_nan(a, b):
    """bitwise equality, except that any NaN equals any NaN (x86 and gfx950 produce default NaNs of opposite sign)"""
    au, bu = a.view(np.uint32), b.view(np.uint32)
    return (au == bu) | (np.isnan(a) & np.isnan(b))


def test_extreme_materials_take_the_ieee_fallbacks(capi):
    """materials whose numbers leave the ranges the short exact forms cover: ior 0 / negative / huge / tiny (Fresnel table with
    inf and NaN entries, refraction with NaN eta), emission that overflows to inf (inf / bounces, inf - inf) or stays
    denormal (denormal / bounces): every variant must still equal the oracle bit for bit, NaNs matching NaNs"""
    import oracle_py as O
    from path_tracer_amd import scene as S
    mats = [S.material(baseColor=(0.8, 0.8, 0.8, 1), emissionColor=(0, 0, 0, 0), smoothness=0.5, ior=v, refractionAmount=0.7, specularChance=0.3)
            for v in (0.0, -1.5, 1e30, 1e-30, 1.0)]
    mats.append(S.material(baseColor=(1, 1, 1, 1), emissionColor=(1, 0.5, 0.25, 1), emissionStrength=3e38))     # overflows when summed
    mats.append(S.material(baseColor=(1, 1, 1, 1), emissionColor=(1, 0.5, 0.25, 1), emissionStrength=1e-42))    # denormal radiance
    mats.append(S.material(baseColor=(0.9, 0.9, 0.9, 1), emissionColor=(0, 0, 0, 0)))                            # ground
    sc = S.Scene(materials=np.stack(mats), mesh=S.bunny_mesh(4))
    sc.add_triangle((-50, 0, 50), (50, 0, 50), (0, 0, -50), 7)
    sc.add_triangle((-1, 0, 3), (1, 0, 3), (0, 1.4, 3), 5)
    sc.add_triangle((-1, 0, -3), (1, 0, -3), (0, 1.4, -3), 6)
    for i in range(7):
        sc.add_sphere((0.9 * np.cos(i), 0.25 + 0.1 * i, 0.9 * np.sin(i)), 0.2, i)
    cam = S.Camera(position=np.array([2.2, 0.8, 0.3], np.float32), yaw=5.0, pitch=12.0).UpdateView()
    W, H, B, SPP = 128, 72, 8, 6
    want = O.Oracle(sc, cam, W, H, B).render(1, SPP)
    assert np.isnan(want).any() or np.isinf(want).any(), "the scene does not reach the non-finite paths"
    for variant in (1, 6, 5):
        got, _ = render(capi, sc, cam, W, H, B, SPP, variant)
        bad = int((~_same_or_both_nan(got, want)).any(-1).sum())
        assert bad == 0, (variant, bad)
    short, _ = render(capi, sc, cam, W, H, B, 2, 6)          # short-launch kernels
    want2 = O.Oracle(sc, cam, W, H, B).render(1, 2)
    assert int((~_same_or_both_nan(short, want2)).any(-1).sum()) == 0
