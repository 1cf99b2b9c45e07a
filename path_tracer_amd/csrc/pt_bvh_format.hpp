// pt_bvh_format.hpp — constants shared by the host BVH builder (pt_bvh.hpp) and the traversal kernels (pt_kernels.hpp).
#pragma once
namespace pt {
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 4
#endif
constexpr int kLeafMax = PT_LEAF_MAX;         // primitives per leaf the builder aims for (the leaf code holds up to 8)
constexpr int kBvhMaxDepth = 40;     // depth bound of one tree; the per-lane traversal stack holds kBvhMaxDepth + 1 entries
// PT_BVH_WIDE = 1: the kernels traverse the four-wide collapse of the builder's binary tree (pt_bvh.hpp BvhNode4, 128 B per
// node); 0: the binary tree itself (BvhNode, 64 B) -- kept for A/B measurements, same leaves and results either way.
#ifndef PT_BVH_WIDE
#define PT_BVH_WIDE 1
#endif
constexpr bool kBvhWide = PT_BVH_WIDE != 0;
constexpr int kBvhNodeVecs = kBvhWide ? 8 : 4;    // float4 per inner node
constexpr float kBvhFarBox = 3.0e38f;             // both corners of an unused child slot: no ray's slab test reaches it (see pt_bvh.hpp)
constexpr int kSphereLeaf = 1 << 30; // leaf-code flag: the leaf holds spheres (records of the sphere tree), not mesh triangles
}  // namespace pt
