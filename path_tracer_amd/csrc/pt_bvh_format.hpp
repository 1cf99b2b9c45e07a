// pt_bvh_format.hpp — constants shared by the host BVH builder (pt_bvh.hpp) and the traversal kernels (pt_kernels.hpp).
#pragma once
namespace pt {
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 4
#endif
constexpr int kLeafMax = PT_LEAF_MAX;         // primitives per leaf the builder aims for (the leaf code holds up to 8)
constexpr int kBvhMaxDepth = 40;     // depth bound of one tree; the per-lane traversal stack holds kBvhMaxDepth + 1 entries
constexpr int kSphereLeaf = 1 << 30; // leaf-code flag: the leaf holds spheres (records of the sphere tree), not mesh triangles
}  // namespace pt
