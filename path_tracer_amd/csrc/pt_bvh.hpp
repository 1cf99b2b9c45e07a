// pt_bvh.hpp — host-side BVH builder for the mesh triangles (binned SAH, BVH2, leaves of <= kLeafMax triangles).
//
// The reference has no working acceleration structure: its shader tests every mesh triangle behind a root-AABB gate
// (rt.comp:213-218; `TraverseBVH`, rt.comp:182-202, is dead code and `Mesh::SplitNode`, src/Object.cpp:183-240, is
// unfinished).  This builder is our own; the traversal kernel (pt_kernels.hpp, bvh_closest) uses it only to SKIP
// triangles that the ray cannot hit, runs the reference's own ray-triangle arithmetic on the rest, and resolves the
// closest hit as the lexicographic minimum of (t, original triangle index) -- which is what the reference's in-order
// `t < closest` scan returns.  Boxes are exact bounds of the vertices; the traversal adds the float safety margins.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/pt_hip.h"
#include "pt_bvh_format.hpp"

namespace pt {

// 64-byte inner node: both child boxes + child codes.  code >= 0: inner node index; code < 0: leaf,
// ~code = flag | first_leaf_primitive << 3 | (count - 1), flag = kSphereLeaf for leaves of the sphere tree.
struct BvhNode {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right, pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode is 4 x float4");

struct Bvh {
    std::vector<BvhNode> nodes;       // nodes[0] is the root
    std::vector<int> order;           // leaf-ordered position -> original triangle index
    int depth = 0;
};

namespace bvh_detail {

struct Box {
    float mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; k++) { mn[k] = INFINITY; mx[k] = -INFINITY; } }
    void grow(const float *p) { for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); } }
    void grow(const Box &b) { for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); } }
    float half_area() const
    {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Ref { Box b; float c[3]; int idx; };

struct Builder {
    std::vector<Ref> refs;
    Bvh *out;
    int leaf_flag = 0;

    static int ceil_log2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

    int make_leaf(int lo, int hi)
    {
        // keep original order inside a leaf (cosmetic: ties are resolved by index, not by visiting order)
        std::sort(refs.begin() + lo, refs.begin() + hi, [](const Ref &a, const Ref &b) { return a.idx < b.idx; });
        const int first = (int)out->order.size();
        for (int i = lo; i < hi; i++) out->order.push_back(refs[i].idx);
        return ~(leaf_flag | (first << 3) | (hi - lo - 1));
    }

    Box bounds(int lo, int hi) const { Box b; b.reset(); for (int i = lo; i < hi; i++) b.grow(refs[i].b); return b; }

    // returns the child code of the subtree over refs[lo, hi)
    int build(int lo, int hi, int depth)
    {
        const int n = hi - lo;
        out->depth = std::max(out->depth, depth);
        if (n <= kLeafMax) return make_leaf(lo, hi);
        Box cb; cb.reset();
        for (int i = lo; i < hi; i++) cb.grow(refs[i].c);
        int axis = 0;
        for (int k = 1; k < 3; k++) if (cb.mx[k] - cb.mn[k] > cb.mx[axis] - cb.mn[axis]) axis = k;
        int mid = -1;
        // SAH only while a median-split subtree below still fits in the depth bound
        const bool may_sah = depth + 1 + ceil_log2((n + kLeafMax - 1) / kLeafMax) + 1 < kBvhMaxDepth;
        if (may_sah && cb.mx[axis] > cb.mn[axis]) {
            constexpr int kBins = 16;
            float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ax++) {
                const float ext = cb.mx[ax] - cb.mn[ax];
                if (!(ext > 0.0f)) continue;
                Box bb[kBins]; int cnt[kBins];
                for (int b = 0; b < kBins; b++) { bb[b].reset(); cnt[b] = 0; }
                const float scale = kBins / ext;
                for (int i = lo; i < hi; i++) {
                    int b = std::min(kBins - 1, std::max(0, (int)((refs[i].c[ax] - cb.mn[ax]) * scale)));
                    bb[b].grow(refs[i].b); cnt[b]++;
                }
                float right_area[kBins]; int right_cnt[kBins];
                Box acc; acc.reset(); int c = 0;
                for (int b = kBins - 1; b > 0; b--) { acc.grow(bb[b]); c += cnt[b]; right_area[b] = c ? acc.half_area() : 0.0f; right_cnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bb[b]); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    const float cost = acc.half_area() * c + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const float ext = cb.mx[best_axis] - cb.mn[best_axis], scale = kBins / ext, mn = cb.mn[best_axis];
                auto it = std::partition(refs.begin() + lo, refs.begin() + hi, [&](const Ref &r) {
                    int b = std::min(kBins - 1, std::max(0, (int)((r.c[best_axis] - mn) * scale)));
                    return b <= best_bin;
                });
                mid = (int)(it - refs.begin());
                if (mid == lo || mid == hi) mid = -1;
            }
        }
        if (mid < 0) {   // object median along the longest centroid axis (also the depth-bounded fallback)
            mid = lo + n / 2;
            std::nth_element(refs.begin() + lo, refs.begin() + mid, refs.begin() + hi,
                             [axis](const Ref &a, const Ref &b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.idx < b.idx); });
        }
        const int me = (int)out->nodes.size();
        out->nodes.emplace_back();
        const Box lb = bounds(lo, mid), rb = bounds(mid, hi);
        const int l = build(lo, mid, depth + 1);
        const int r = build(mid, hi, depth + 1);
        BvhNode &nd = out->nodes[me];
        for (int k = 0; k < 3; k++) { nd.lmin[k] = lb.mn[k]; nd.lmax[k] = lb.mx[k]; nd.rmin[k] = rb.mn[k]; nd.rmax[k] = rb.mx[k]; }
        nd.left = l; nd.right = r; nd.pad0 = nd.pad1 = 0;
        return me;
    }
};

}  // namespace bvh_detail

// Builds over refs already filled with boxes/centroids/indices.
inline void build_from_refs(bvh_detail::Builder &b, Bvh &out)
{
    using namespace bvh_detail;
    const int n = (int)b.refs.size();
    out.nodes.clear(); out.order.clear(); out.depth = 0;
    if (n <= 0) return;
    b.out = &out;
    out.order.reserve(n);
    out.nodes.reserve(std::max(1, n / 2));
    if (n <= kLeafMax) {
        // the traversal starts at an inner node: root = (leaf of everything, the same leaf again)
        out.nodes.emplace_back();
        const Box lb = b.bounds(0, n);
        const int l = b.make_leaf(0, n);
        BvhNode &nd = out.nodes[0];
        // the right child repeats the left one: visiting a leaf twice cannot change the (t, index) minimum
        for (int k = 0; k < 3; k++) { nd.lmin[k] = nd.rmin[k] = lb.mn[k]; nd.lmax[k] = nd.rmax[k] = lb.mx[k]; }
        nd.left = l; nd.right = l; nd.pad0 = nd.pad1 = 0;
        out.depth = 1;
        return;
    }
    b.build(0, n, 0);
}

inline void build_bvh(const PtTriangle *tris, int n, Bvh &out)
{
    using namespace bvh_detail;
    Builder b; b.refs.resize(std::max(0, n));
    for (int i = 0; i < n; i++) {
        Ref &r = b.refs[i];
        r.b.reset(); r.b.grow(tris[i].p1); r.b.grow(tris[i].p2); r.b.grow(tris[i].p3);
        for (int k = 0; k < 3; k++) r.c[k] = 0.5f * r.b.mn[k] + 0.5f * r.b.mx[k];
        r.idx = i;
    }
    build_from_refs(b, out);
}

// The spheres of the scene (SSBO 4): boxes centre -+ radius (the traversal's per-ray margin covers the rounding).
inline void build_sphere_bvh(const PtSphere *sph, int n, Bvh &out)
{
    using namespace bvh_detail;
    Builder b; b.refs.resize(std::max(0, n)); b.leaf_flag = kSphereLeaf;
    for (int i = 0; i < n; i++) {
        Ref &r = b.refs[i];
        const float rad = std::fabs(sph[i].radius);
        for (int k = 0; k < 3; k++) { r.b.mn[k] = sph[i].position[k] - rad; r.b.mx[k] = sph[i].position[k] + rad; r.c[k] = sph[i].position[k]; }
        r.idx = i;
    }
    build_from_refs(b, out);
}

}  // namespace pt
