// pt_bvh.hpp — host-side BVH builder for the mesh triangles (binned SAH, BVH2, leaves of <= kLeafMax triangles, collapsed to a four-wide tree).
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

// 128-byte four-wide inner node, what the traversal kernels read (PT_BVH_WIDE): the boxes of up to four children, component by
// component (so that child k is lane k of six float4 loads), and their codes (same encoding as BvhNode's).  An unused slot
// holds a box far outside every scene -- both corners at (kBvhFarBox, kBvhFarBox, kBvhFarBox); uploads refuse coordinates beyond
// 1e18 -- so the slab test rejects it like any box the ray misses, without a test of its own: per axis both planes give the same
// t, either > every hit distance or < 0.  (An "empty" box with min > max would NOT do: the test sorts each axis' pair.)  Its
// code is kBvhNoChild.  One node = one 128-byte cache line
// where the tree lives in global memory: a node visit is ONE dependent fetch for four children instead of two for two each.
struct BvhNode4 {
    float mnx[4], mny[4], mnz[4], mxx[4], mxy[4], mxz[4];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(BvhNode4) == 128, "BvhNode4 is 8 x float4");
constexpr int32_t kBvhNoChild = (int32_t)0x80000000;      // == the traversal's kTravDone: never followed (its box is empty)

// The same node in 64 bytes, for trees that live in global memory (traversal variant 5): child boxes as 8-bit offsets from the
// node's own corner `o`, in steps of 2^e per axis.  A node visit is then FOUR 16-byte loads per lane instead of seven -- the
// vector-memory pipeline, not arithmetic, bounds the traversal of a tree that does not fit in LDS -- and the
// whole tree is half as large in L2.  Quantisation is conservative: the box a lane reconstructs, fmaf(q, 2^e, o) per plane --
// the very operation quantise_bvh4 checks with -- contains the exact box; boxes only ever decide what may be SKIPPED.
//   float4 0: o.x, o.y, o.z, bits(ex | ey << 8 | ez << 16)   (e as the biased exponent byte of the float 2^e)
//   float4 1: child codes
//   float4 2: qmin.x of children 0..3 (one byte each) | qmin.y | qmin.z | qmax.x      float4 3: qmax.y | qmax.z | - | -
struct BvhNode4Q {
    float o[3];
    uint32_t exps;
    int32_t child[4];
    uint8_t qmn[3][4], qmx[3][4];
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode4Q) == 64, "BvhNode4Q is 4 x float4");

struct Bvh4 {
    std::vector<BvhNode4> nodes;      // nodes[0] is the root
    int stack_need = 0;               // most entries one lane's stack can hold at once (every visit may push all but one child)
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
    int leaf_max = kLeafMax;          // primitives per leaf the builder aims for

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
        if (n <= leaf_max) return make_leaf(lo, hi);
        Box cb; cb.reset();
        for (int i = lo; i < hi; i++) cb.grow(refs[i].c);
        int axis = 0;
        for (int k = 1; k < 3; k++) if (cb.mx[k] - cb.mn[k] > cb.mx[axis] - cb.mn[axis]) axis = k;
        int mid = -1;
        // SAH only while a median-split subtree below still fits in the depth bound
        const bool may_sah = depth + 1 + ceil_log2((n + leaf_max - 1) / leaf_max) + 1 < kBvhMaxDepth;
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
    if (n <= b.leaf_max) {
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

// Collapse the binary tree into a four-wide one: a node starts with the two children of a binary node and, while it has a free
// slot, replaces the inner child with the largest box by that child's own two children.  Leaves, their order and every box are
// the binary tree's: the set of triangles a ray can reach is unchanged, only the number of steps to reach them is.
// `base` is added to every inner child index (the sphere tree's nodes follow the mesh tree's in one array).
inline void collapse_bvh4(const std::vector<BvhNode> &bin, Bvh4 &out, int base = 0)
{
    using bvh_detail::Box;
    out.nodes.clear(); out.stack_need = 0; out.depth = 0;
    if (bin.empty()) return;
    struct Slot { Box b; int code; };
    struct Rec {
        const std::vector<BvhNode> &bin; Bvh4 &out; int base;
        void kids(const BvhNode &n, Slot &l, Slot &r) const
        {
            for (int k = 0; k < 3; k++) { l.b.mn[k] = n.lmin[k]; l.b.mx[k] = n.lmax[k]; r.b.mn[k] = n.rmin[k]; r.b.mx[k] = n.rmax[k]; }
            l.code = n.left; r.code = n.right;
        }
        // returns the wide node's index; need = stack entries a traversal below it can hold, depth = levels below it
        int make(int bi, int depth, int &need)
        {
            Slot s[4]; int ns = 2;
            kids(bin[bi], s[0], s[1]);
            while (ns < 4) {
                int pick = -1; float area = -1.0f;
                for (int i = 0; i < ns; i++) if (s[i].code >= 0 && s[i].b.half_area() > area) { area = s[i].b.half_area(); pick = i; }
                if (pick < 0) break;
                Slot l, r;
                kids(bin[s[pick].code], l, r);
                s[pick] = l; s[ns++] = r;
            }
            const int me = (int)out.nodes.size();
            out.nodes.emplace_back();
            out.depth = std::max(out.depth, depth + 1);
            int below = 0, codes[4];
            for (int i = 0; i < ns; i++) {
                codes[i] = s[i].code;
                if (s[i].code >= 0) { int nd = 0; codes[i] = make(s[i].code, depth + 1, nd) + base; below = std::max(below, nd); }
            }
            BvhNode4 &w = out.nodes[me];
            for (int i = 0; i < 4; i++) {
                const bool used = i < ns;
                w.mnx[i] = used ? s[i].b.mn[0] : kBvhFarBox; w.mny[i] = used ? s[i].b.mn[1] : kBvhFarBox; w.mnz[i] = used ? s[i].b.mn[2] : kBvhFarBox;
                w.mxx[i] = used ? s[i].b.mx[0] : kBvhFarBox; w.mxy[i] = used ? s[i].b.mx[1] : kBvhFarBox; w.mxz[i] = used ? s[i].b.mx[2] : kBvhFarBox;
                w.child[i] = used ? codes[i] : kBvhNoChild; w.pad[i] = 0;
            }
            need = (ns - 1) + below;       // this visit leaves at most ns - 1 siblings on the stack while the traversal goes on below
            return me;
        }
    } rec{bin, out, base};
    int need = 0;
    rec.make(0, 0, need);
    out.stack_need = need;
}

// 64-byte form of a four-wide node; returns false if a box cannot be represented (never for finite boxes within +-1e18)
inline bool quantise_bvh4(const BvhNode4 &w, BvhNode4Q &q)
{
    std::memset(&q, 0, sizeof q);
    const float *mn[3] = {w.mnx, w.mny, w.mnz}, *mx[3] = {w.mxx, w.mxy, w.mxz};
    for (int k = 0; k < 3; k++) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < 4; i++) if (w.child[i] != kBvhNoChild) { lo = std::min(lo, mn[k][i]); hi = std::max(hi, mx[k][i]); }
        if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) return false;
        // smallest step 2^e with (hi - lo) / 2^e <= 253: two codes stay free for the outward corrections below
        int e = -126;
        const double ext = (double)hi - (double)lo;
        if (ext > 0.0) { int ex; std::frexp(ext / 253.0, &ex); e = std::max(-126, ex); }      // 2^ex > ext / 253 >= 2^(ex-1)
        if (e > 127) return false;
        const float step = std::ldexp(1.0f, e);
        q.o[k] = lo;
        q.exps |= (uint32_t)(e + 127) << (8 * k);
        for (int i = 0; i < 4; i++) {
            if (w.child[i] == kBvhNoChild) { q.qmn[k][i] = 255; q.qmx[k][i] = 0; continue; }
            int a = (int)std::floor(((double)mn[k][i] - (double)lo) / (double)step);
            int b = (int)std::ceil(((double)mx[k][i] - (double)lo) / (double)step);
            a = std::min(std::max(a, 0), 255); b = std::min(std::max(b, 0), 255);
            // the lane evaluates fmaf(q, step, o) in float: step the codes outwards until THAT value encloses the exact plane
            while (a > 0 && std::fmaf((float)a, step, lo) > mn[k][i]) a--;
            while (b < 255 && std::fmaf((float)b, step, lo) < mx[k][i]) b++;
            if (std::fmaf((float)a, step, lo) > mn[k][i] || std::fmaf((float)b, step, lo) < mx[k][i]) return false;
            q.qmn[k][i] = (uint8_t)a; q.qmx[k][i] = (uint8_t)b;
        }
    }
    for (int i = 0; i < 4; i++) q.child[i] = w.child[i];
    return true;
}

// What the float evaluation of rt.comp:153-173 makes of a mesh triangle, as far as a bounding box is concerned.  The test
// divides by det = e1 . (d x e2) = |d| |e1 x e2| cos(phi) and scales u, v and t by the quotient: a relative error of
// c eps |e1||e2| / (|e1 x e2| cos(phi)) in det moves the accepted region in the triangle's plane by that fraction of the
// triangle's size (measured: tools/fuzz_bvh.py, profiles/r3_exactness).  The box of a triangle is therefore grown by
// kSliverGrow * (longest edge) / sin(angle at p1) on every side -- 0.05 % of its size for a well-shaped triangle, which keeps
// the tree exact for rays down to ~1e-3 rad from the triangle's plane, and in proportion more for slivers.  Where that growth
// exceeds the mesh itself (edges parallel within rounding: det is noise and the shader reports hits anywhere along the
// triangle's line) the triangle goes on a list that every ray behind the root-AABB gate tests, like the shader does; a
// triangle with a zero edge at p1 can never be hit (det = +-0, t = inf * 0 = NaN fails `t > HIT_LIMIT`) and is dropped.
constexpr double kSliverGrow = 5e-4;
enum TriClass { kTriTree = 0, kTriListed = 1, kTriNever = 2 };
// e1 = p2 - p1, e2 = p3 - p1 as the kernels hold them (rounded to float); extent: largest |coordinate| of the mesh
inline TriClass classify_triangle(const float e1[3], const float e2[3], double extent, float &grow)
{
    const double a[3] = {e1[0], e1[1], e1[2]}, b[3] = {e2[0], e2[1], e2[2]};
    const double l1 = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), l2 = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    grow = 0.0f;
    if (!PT_EXACT_MARGINS) return kTriTree;
    if (l1 == 0.0 || l2 == 0.0) return kTriNever;
    const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
    const double cr = std::sqrt(cx * cx + cy * cy + cz * cz);
    const double l3 = std::sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]) + (b[2] - a[2]) * (b[2] - a[2]));
    const double r = cr > 0.0 ? kSliverGrow * std::max(l1, std::max(l2, l3)) * (l1 * l2) / cr : INFINITY;
    if (!(r <= 4.0 * extent)) return kTriListed;
    grow = (float)r;
    return kTriTree;
}

// The triangles no box can stand for (kTriListed) are given leaves whose boxes every ray meets, under a new root beside the
// old one (whose box is the tree's own bounds): a ray behind the root-AABB gate tests all of them, like the shader, and the
// traversal kernels need no code for it.  kBvhEverything is the largest coordinate the builder accepts (1e18 at upload), so
// the slab tests' products stay finite (pt_kernels.hpp slab_inv).  Leaf order: the listed triangles follow the tree's.
constexpr float kBvhEverything = 9.0e17f;
inline void attach_listed(Bvh &bvh, const std::vector<int> &listed)
{
    if (listed.empty() || bvh.nodes.empty()) return;
    std::vector<int> leaves;
    for (size_t i = 0; i < listed.size(); i += kLeafMax) {
        const int first = (int)bvh.order.size(), cnt = (int)std::min<size_t>(kLeafMax, listed.size() - i);
        for (int k = 0; k < cnt; k++) bvh.order.push_back(listed[i + k]);
        leaves.push_back(~((first << 3) | (cnt - 1)));
    }
    std::vector<BvhNode> top(1);       // top[0]: the new root, filled in last
    int depth = 0;
    struct Rec {
        std::vector<BvhNode> &top; const std::vector<int> &leaves; int &depth;
        int make(int lo, int hi, int d)       // child code of a balanced subtree over leaves[lo, hi); inner nodes index into `top`
        {
            depth = std::max(depth, d);
            if (hi - lo == 1) return leaves[lo];
            const int me = (int)top.size();
            top.emplace_back();
            const int mid = (lo + hi) / 2, l = make(lo, mid, d + 1), r = make(mid, hi, d + 1);
            BvhNode &nd = top[me];
            for (int k = 0; k < 3; k++) { nd.lmin[k] = nd.rmin[k] = -kBvhEverything; nd.lmax[k] = nd.rmax[k] = kBvhEverything; }
            nd.left = l; nd.right = r; nd.pad0 = nd.pad1 = 0;
            return me;
        }
    } rec{top, leaves, depth};
    const int sub = rec.make(0, (int)leaves.size(), 1);
    const int shift = (int)top.size();
    BvhNode root;
    const BvhNode &old = bvh.nodes[0];
    for (int k = 0; k < 3; k++) {
        root.lmin[k] = -kBvhEverything; root.lmax[k] = kBvhEverything;
        root.rmin[k] = std::min(old.lmin[k], old.rmin[k]); root.rmax[k] = std::max(old.lmax[k], old.rmax[k]);
    }
    root.left = sub; root.right = shift; root.pad0 = root.pad1 = 0;
    top[0] = root;
    for (BvhNode nd : bvh.nodes) {
        if (nd.left >= 0) nd.left += shift;
        if (nd.right >= 0) nd.right += shift;
        top.push_back(nd);
    }
    bvh.nodes.swap(top);
    bvh.depth = std::max(bvh.depth + 1, depth + 1);
}

// grow / cls (optional, per triangle): box growth and class from classify_triangle; only kTriTree triangles enter the tree
inline void build_bvh(const PtTriangle *tris, int n, Bvh &out, const float *grow = nullptr, const unsigned char *cls = nullptr)
{
    using namespace bvh_detail;
    Builder b; b.refs.reserve(std::max(0, n));
    for (int i = 0; i < n; i++) {
        if (cls && cls[i] != kTriTree) continue;
        Ref r;
        r.b.reset(); r.b.grow(tris[i].p1); r.b.grow(tris[i].p2); r.b.grow(tris[i].p3);
        for (int k = 0; k < 3; k++) r.c[k] = 0.5f * r.b.mn[k] + 0.5f * r.b.mx[k];
        if (grow) for (int k = 0; k < 3; k++) { r.b.mn[k] -= grow[i]; r.b.mx[k] += grow[i]; }
        r.idx = i;
        b.refs.push_back(r);
    }
    build_from_refs(b, out);
}

// The spheres of the scene (SSBO 4): boxes centre -+ radius (the traversal's per-ray margin covers the rounding).
inline void build_sphere_bvh(const PtSphere *sph, int n, Bvh &out)
{
    using namespace bvh_detail;
    Builder b; b.refs.resize(std::max(0, n)); b.leaf_flag = kSphereLeaf; b.leaf_max = kSphLeafMax;
    for (int i = 0; i < n; i++) {
        Ref &r = b.refs[i];
        const float rad = std::fabs(sph[i].radius);
        for (int k = 0; k < 3; k++) { r.b.mn[k] = sph[i].position[k] - rad; r.b.mx[k] = sph[i].position[k] + rad; r.c[k] = sph[i].position[k]; }
        r.idx = i;
    }
    build_from_refs(b, out);
}

}  // namespace pt
