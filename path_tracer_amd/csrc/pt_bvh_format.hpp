// pt_bvh_format.hpp — constants shared by the host BVH builder (pt_bvh.hpp) and the traversal kernels (pt_kernels.hpp).
#pragma once
namespace pt {
// Primitives per leaf the builder aims for (a leaf code holds up to 8).  With the four-wide tree the optimum is 2 triangles -- one
// leaf step tests two -- and 1 sphere: a box test is cheap next to a primitive test, and four boxes are tested per node visit.
// Measured (kernel ms, configs 2 / 3 / 5; profiles/r3_bvh4/ab_triangle_leaf.txt, ab_sphere_leaf.txt): triangles per leaf
// 6: 6.14 / 30.3 / 31.7, 4: 6.01 / 28.7 / 30.6, 3: 6.00 / 26.9 / 30.1, 2: 5.85 / 26.9 / 29.0, 1: 6.14 / 31.0 / 36.6;
// spheres per leaf (config 5, 4 triangles per leaf) 4: 35.2, 3: 35.2, 2: 32.2, 1: 30.7.
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 2
#endif
#ifndef PT_SPH_LEAF_MAX
#define PT_SPH_LEAF_MAX 1
#endif
constexpr int kLeafMax = PT_LEAF_MAX;
constexpr int kSphLeafMax = PT_SPH_LEAF_MAX;
constexpr bool kOneStepLeaves = kLeafMax <= 2 && kSphLeafMax <= 1;      // every leaf is used up by ONE leaf step (two triangles or one sphere)
constexpr int kBvhMaxDepth = 40;     // depth bound of one tree; the per-lane traversal stack holds kBvhMaxDepth + 1 entries
// PT_BVH_WIDE = 1: the kernels traverse the four-wide collapse of the builder's binary tree (pt_bvh.hpp BvhNode4, 128 B per
// node); 0: the binary tree itself (BvhNode, 64 B) -- kept for A/B measurements, same leaves and results either way.
#ifndef PT_BVH_WIDE
#define PT_BVH_WIDE 1
#endif
constexpr bool kBvhWide = PT_BVH_WIDE != 0;
constexpr int kBvhNodeVecs = kBvhWide ? 8 : 4;    // float4 per inner node in global memory
// ... and in the LDS copy: the wide node's eighth float4 is padding.  Leaving it out saves an eighth of the copy and moves the
// node stride from 32 banks to 28 (112 B): node starts then fall on 16 different banks instead of 2, which matters for a
// gather in which every lane reads its own node
constexpr int kBvhLdsNodeVecs = kBvhWide ? 7 : 4;
constexpr int kBvhTriVecs = 4;                    // float4 per leaf-ordered triangle record in global memory: p1|index, e1|material, e2, normal
// mesh triangles no bounding box can stand for (pt_bvh.hpp classify_triangle) sit in leaves every ray reaches; more of them than
// this AND than half of the mesh, and the mesh is traced by the brute-force kernels instead
constexpr int kMeshListMax = 64;
// rt.comp:126-146 accepts rays that pass a sphere of radius r at a distance up to sqrt(r^2 + k |o - c|^2) from its centre, k ~ 10 eps
// (rounding of hb^2 - a c); the traversal's margin uses 16 eps
#ifndef PT_SLAB_BY_SIGN
#define PT_SLAB_BY_SIGN 1          // the wide node step reads a node's planes sorted by the ray's direction (0: min / max per axis)
#endif
#ifndef PT_EXACT_MARGINS
#define PT_EXACT_MARGINS 1        // 0: A/B builds without the triangle classes and the sphere margin (not exact for slivers / far spheres)
#endif
constexpr float kSphereReachK = PT_EXACT_MARGINS ? 1.0e-6f : 0.0f;
constexpr float kBvhFarBox = 3.0e38f;             // both corners of an unused child slot: no ray's slab test reaches it (see pt_bvh.hpp)
constexpr int kSphereLeaf = 1 << 30; // leaf-code flag: the leaf holds spheres (records of the sphere tree), not mesh triangles
}  // namespace pt
