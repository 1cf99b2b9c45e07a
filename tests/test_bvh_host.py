"""Host-side BVH builder of libpt_hip.so (path_tracer_amd/csrc/pt_bvh.hpp), exercised without a GPU through
pt_bvh_selftest: structure invariants the traversal kernel relies on."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def capi():
    from path_tracer_amd import capi as c
    c.load()
    return c


def test_bunny(capi):
    from path_tracer_amd import scene as S
    m = S.bunny_mesh(3)
    rc, nodes, depth, max_leaf, sah = capi.bvh_selftest(m.tris)
    assert rc == 0
    assert 1 <= max_leaf <= 2 and len(m.tris) / 2 <= nodes + 1 <= len(m.tris)      # two triangles per leaf (pt_bvh_format.hpp)
    assert depth <= 24, "a 656-triangle mesh must not come near the 40-entry traversal stack"
    assert 0 < sah < 60


@pytest.mark.parametrize("n", [1, 2, 4, 5, 9])
def test_tiny(capi, n):
    from path_tracer_amd import scene as S
    t = S.heightfield_mesh(4, 1).tris[:n]
    rc, nodes, depth, max_leaf, _ = capi.bvh_selftest(t)
    assert rc == 0 and nodes >= 1 and max_leaf <= 4


def test_degenerate_inputs_keep_the_depth_bound(capi):
    from path_tracer_amd import scene as S
    # 5000 copies of one triangle: no spatial split exists, the builder must fall back to median splits
    one = S.heightfield_mesh(1, 1).tris[:1]
    rc, nodes, depth, max_leaf, _ = capi.bvh_selftest(np.repeat(one, 5000))
    assert rc == 0 and depth <= 40 and max_leaf <= 4
    # a long sliver row (all centroids on one line, exponentially growing sizes): SAH makes deep trees, the bound holds
    t = np.repeat(one, 3000)
    for k, name in enumerate(("p1", "p2", "p3")):
        t[name][:, 0] = (1.001 ** np.arange(3000) + 0.01 * k).astype(np.float32)
        t[name][:, 1] = 0.1 * k
        t[name][:, 2] = 0.0
    rc, nodes, depth, max_leaf, _ = capi.bvh_selftest(t)
    assert rc == 0 and depth <= 40


def test_one_million_triangles(capi):
    from path_tracer_amd import scene as S
    m = S.heightfield_mesh(708, 3)
    rc, nodes, depth, max_leaf, sah = capi.bvh_selftest(m.tris)
    assert rc == 0 and len(m.tris) == 1002528 and depth <= 40 and max_leaf <= 4


def test_non_finite_is_refused(capi):
    from path_tracer_amd import scene as S
    t = S.heightfield_mesh(4, 1).tris.copy()
    t["p3"][7][2] = np.inf
    assert capi.bvh_selftest(t)[0] == capi.PT_E_INVALID


def test_triangle_classes_for_the_tree(capi):
    """pt_classify_mesh (pt_bvh.hpp classify_triangle): what the shader's float triangle test can report outside a triangle
    decides how the tree boxes it -- a small growth for well-shaped triangles, a large one for slivers, the always-tested list
    where the edges at p1 are parallel within rounding, nothing for triangles with a zero edge at p1 (never hit)."""
    from path_tracer_amd import scene as S
    bunny = S.bunny_mesh(3).tris
    rc, tree, listed, never, grow = capi.classify_mesh(bunny)
    assert rc == 0 and (tree, listed, never) == (len(bunny), 0, 0)
    ext = float(np.abs(np.concatenate([bunny["p1"], bunny["p2"], bunny["p3"]])).max())
    assert 0 < grow < 0.02 * ext, "bunny1 has no sliver whose box would grow by more than 2 % of the mesh"
    pts = np.array([
        [[0, 0, 0], [1, 0, 0], [0, 1, 0]],            # right angle at p1: growth 5e-4 x sqrt(2)
        [[0, 0, 0], [1, 0, 0], [2, 0, 0]],            # zero area, edges parallel: listed
        [[0, 0, 0], [1, 0, 0], [1, 0, 0]],            # p2 == p3: listed
        [[0, 0, 0], [0, 0, 0], [1, 1, 0]],            # p1 == p2: zero edge at p1, never hit
        [[1, 1, 1], [1, 1, 1], [1, 1, 1]],            # a point: never hit
        [[0, 0, 0], [1, 0, 0], [1, 1e-3, 0]],         # sliver, sin(angle at p1) = 1e-3: growth 0.5 x its length, still in the tree
        [[0, 0, 0], [1, 0, 0], [1, 1e-5, 0]],         # sin = 1e-5: growth 50 x its length > 4 x the extent: listed
    ], np.float32)
    t = S.triangles_from_points(pts, 0)
    rc, tree, listed, never, grow = capi.classify_mesh(t)
    assert rc == 0 and (tree, listed, never) == (2, 3, 2)
    assert abs(grow - 0.5) < 0.01
    rc, tree, listed, never, grow = capi.classify_mesh(t[:1])
    assert (tree, listed, never) == (1, 0, 0) and abs(grow - 5e-4 * np.sqrt(2.0)) < 1e-6
    bad = t[:1].copy(); bad["p2"][0, 0] = np.nan
    assert capi.classify_mesh(bad)[0] == capi.PT_E_INVALID
