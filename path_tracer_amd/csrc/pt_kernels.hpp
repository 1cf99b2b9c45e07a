// pt_kernels.hpp — HIP kernels of the path-tracing hot path for gfx950 (MI355X, wave64).
//
// Two families of trace kernels, same work decomposition (persistent waves, work items = a tile of 64 px (tile_pixel: an 8 x 8 block) x a range of
// passes, pulled from an atomic queue in heaviest-first order), same per-sample arithmetic (pt_math.hpp):
//
//  * pt_trace_items_bvh_lds / pt_trace_items_bvh (default): every lane runs a small path state machine; rays that pass the
//    root-AABB gate are parked and traversed together through a BVH (pt_bvh.hpp) that only SKIPS triangles -- the tests
//    that remain are the reference's, the winner is the lexicographic minimum of (t, canonical id).  See the comment
//    above trace_items_body_bvh.
//
//  * pt_trace_items_lds / pt_trace_items ("triangle-stationary waves", exact by construction): pixel work runs one pixel
//    per lane, but the brute-force ray/triangle loop against the mesh (rt.comp:213-218) runs the other way round: every
//    lane keeps SLOTS mesh triangles (in a workgroup-shared LDS image, or in VGPRs), and the rays that passed the gate are
//    taken one at a time from the wave's ballot mask, broadcast to all lanes (v_readlane), tested against 64*SLOTS
//    triangles at once, and reduced with a DPP min + ballot.  Cost is proportional to the number of rays that need the
//    mesh, however they are scattered over lanes.  Ties resolve to the lowest primitive index exactly like the shader's
//    in-order strict `<` scan: lane l holds triangles [l*SLOTS, (l+1)*SLOTS), so "lowest lane among equal t" is "lowest
//    index".
#pragma once
#include "pt_math.hpp"
#include "pt_bvh_format.hpp"
#include "../../include/pt_hip.h"

#pragma clang fp contract(off)

namespace pt {

#ifndef PT_SLOT_GROUP
#define PT_SLOT_GROUP 1      // triangle slots the instruction scheduler may interleave
#endif
#ifndef PT_SCHED_MASK
#define PT_SCHED_MASK 0      // what may cross a slot boundary (0: nothing; 0x100: DS reads, so LDS prefetch can run ahead)
#endif
constexpr int kWave = 64;
constexpr int kSlots = 11;                 // mesh triangles per lane held in registers
constexpr int kNumCounters = 18;
constexpr int kBvhStack = kBvhMaxDepth + 1;   // per-lane traversal stack entries (one extra: the sphere tree's root under the mesh tree)
constexpr int kTravDone = (int)0x80000000;      // traversal position of a lane that is not traversing; the bottom of every stack; the code of an unused child slot
constexpr int kChunk = kWave * kSlots;     // 704 triangles per register-resident chunk (bunny1: 656)

// PtShadingParams (include/pt_hip.h: the literals in which rt.frag differs from rt.comp) as the kernels take them:
// wave-uniform kernel arguments, filled by shading_of() in pt_api.hip.
struct Shading {
    float hit_limit;          // HIT_LIMIT (rt.comp:10)
    float origin_offset;      // rt.comp:305
    float jitter_scale;       // rt.comp:332
    float sky[3];             // skyColor (rt.comp:25): multiplies rayColor at a miss (rt.comp:316)
    float sky_emit[3];        // skyColor * skyIntensity (rt.comp:315), multiplied on the host in float32
    float box_slack;          // traversal: a box whose exit lies before this (negative) t holds nothing that counts as hit
    float pix_off;            // 0: integer texel coordinates (rt.comp:12) | 0.5: pixel centres (rt.frag:14)
    int flags;                // kSh* bits: where the two shaders differ in the form of an expression, not in a literal
};
enum : int {
    kShN1One = 1,             // FresnelReflectAmount's n1 = 1.0 (rt.frag:283) instead of ior (rt.comp:283)
    kShTwoDraws = 2,          // second RandomInHemisphere for the refraction branch (rt.frag:295)
    kShSqrtOut = 4,           // sqrt(incoming / bounces) (rt.frag:320)
    kShSumXyz = 8,            // sphere test: squared lengths summed (x*x + y*y) + z*z (rt.frag:89-91)
    kShTraceMesh = 16,        // the mesh is traced behind the root-AABB gate (rt.comp:213-218; not in rt.frag:207-216)
    kShDefault = kShTraceMesh // rt.comp
};
constexpr float kBoxSlack = -1e-4f;   // Shading::box_slack for rt.comp's hit limit

struct SceneView {
    Shading sh;               // how it is to be shaded (not scene data; here because every intersection routine takes the view)
    const float *mesh_soa;    // [n_chunks][9][kChunk]; component c of triangle (chunk, lane, slot) at c*kChunk + slot*64 + lane
    const float4 *mesh_p1;    // per mesh triangle: xyz = p1, w = bits(materialIndex)
    const float4 *mesh_e1;    // xyz = p2 - p1
    const float4 *mesh_e2;    // xyz = p3 - p1
    const float4 *mesh_nrm;   // xyz = normalize(cross(e2, e1))   (rt.comp:171)
    int n_mesh, n_chunks;
    float bmin[3], bmax[3];   // nodes[0] (rt.comp:213)
    const float4 *tri_p1, *tri_e1, *tri_e2, *tri_nrm;   // scene triangles (SSBO 5), same encoding
    int n_tris;
    const float4 *sph;        // xyz = centre, w = radius (SSBO 4)
    const uint32_t *sph_mat;
    int n_sph;
    const PtMaterial *mats;   // SSBO 3
    const float4 *prim_tab;   // [n_mesh + n_tris][4]: (p1 | material bits, e1, e2, normal) by canonical id -- the traversal kernels' only triangle table
    const float4 *mat_fres;   // per material, computed on the host with the same IEEE operations: 1/ior, r0 and n1/n2 of a front-face hit (pt_math.hpp fresnel_reflect_amount)
    // BVH over the mesh triangles (pt_bvh.hpp), used by the traversal variant
    const float4 *bvh_nodes;  // kBvhNodeVecs x float4 per inner node: child boxes + child codes (pt_bvh.hpp BvhNode4 / BvhNode); staged into LDS by variant 6
    const float4 *bvh_nodes_q; // the four-wide nodes in their 64-byte quantised form (BvhNode4Q), read in place by variant 5
    const float4 *bvh_tris;   // kBvhTriVecs x float4 per triangle in leaf order: (p1, bits(original index)), (e1, material), e2, normal
    float bvh_scale;          // largest |coordinate| of the mesh and the spheres (for the traversal's safety margin)
    const float4 *bvh_sph;    // sphere tree leaves, 2 x float4 per sphere in leaf order: (centre, radius), (bits(sphere index), -, -, -)
    float sph_bmin[3], sph_bmax[3];   // bounds of all spheres (centre -+ |radius|); read once per workgroup (stage_scene_lds)
    int sph_root;             // node index of the sphere tree's root in bvh_nodes, -1: spheres are scanned one by one
    int n_bvh_tris;           // triangles in the tree (= records in bvh_tris): n_mesh minus the ones the shader can never hit
    // Spheres seen from far away: rt.comp:126-146 reports hits up to sqrt(r^2 + k |o - c|^2) - r outside the sphere; the sphere
    // tree's slab tests get that much more margin per ray, from the smallest radius and the farthest centre (|.|_1)
    float sph_k, sph_rmin, sph_reach;     // the trace kernels read them from LDS too
};

// one sample's radiance in the workspace: three floats (the alpha the shader stores is the constant 1)
struct __attribute__((aligned(4))) Rgb { float r, g, b; };

struct FrameParams {
    int width, height;
    float cam[3];
    float iv[9];              // inverseView columns 0,1,2 (xyz each); column 3 and row w are never read (rt.comp:333)
    int max_bounces, debug_normal;
    int first_pass, n_passes; // passes of this batch: first_pass .. first_pass + n_passes - 1
    int chunk;                // passes per work item of the mesh tiles (class 0)
    int traced_width, rows, x_blocks;   // (W/8)*8, (H/8)*8 (glDispatchCompute(W/8, H/8, 1)), tiles per row of tiles (tile_pixel)
    int sum_mode;             // 0: running mean (rt.comp:337), 1: plain sum for the multi-GPU split
    float *accum;             // RGBA32F, row 0 = bottom
    Rgb *radiance;            // [n_passes][H][W] 1-spp radiance of this batch, 12 B per sample
    unsigned *queue;          // work-item counter, zeroed before every launch
    // schedule (pt_tile_weights + pt_sort_tiles): tiles = 64 px (tile_pixel), sorted heaviest first, in three classes
    const unsigned *order;    // [n_tiles] tile index by rank
    const int *class_count;   // [3] tiles per class: 0 = some pixel passes the mesh gate, 1 = hits only scene primitives, 2 = sky
    int class_ppi[3];         // passes per work item for each class
    int per_tile[3];          // work items per tile for each class = ceil(n_passes / class_ppi) (host-computed)
    unsigned *weights;        // per-tile sort key, written by pt_tile_weights
    int park_high, park_wait; // BVH body: traverse when this many lanes are parked, or after this many iterations of waiting
    int refill_min;           // BVH body: hand out new pixels once this many lanes are idle
    int park_exit;            // BVH body: leave the traversal phase when park_exit/8 of its rays are still unfinished
    int n_tiles;
    unsigned long long *counters;   // [0] samples [1] segments [2] mesh segments [3] mesh tri tests [4] shaded hits [5] wave iterations
                                    // [6] primary mesh segments [7] waves [8] sum of wave spans (ticks) [9] longest wave span [10] BVH node visits [11] BVH triangle tests
                                    // [12..15] phase cycles (diagnostic builds) [16] sphere tests in sphere-tree leaves [17] BVH box tests
    // primary-hit probe outputs (test-only kernel)
    int32_t *out_ids; float *out_t; float *out_dirs; uint32_t *out_rng;
};

// Tile = the 64 pixels one work item covers.  PT_TILE_8X8: an 8 x 8 block (the reference's own workgroup, rt.comp:3: the
// traced area is a whole number of them) -- the rays of a wave stay close together, so they visit the same nodes;
// otherwise 64 pixels of one row.  x_blocks = tiles per row of tiles.  Returns false for a pixel outside the traced area.
#ifndef PT_TILE_8X8
#define PT_TILE_8X8 1
#endif
__device__ __forceinline__ bool tile_pixel(const FrameParams &fp, int tile, int r, int &x, int &y)
{
    const int bx = tile % fp.x_blocks, by = tile / fp.x_blocks;
#if PT_TILE_8X8
    x = bx * 8 + (r & 7); y = by * 8 + (r >> 3);
    return true;
#else
    x = bx * kWave + r; y = by;
    return x < fp.traced_width;
#endif
}
__device__ __forceinline__ int tile_pixels(const FrameParams &fp, int tile)      // pixels of the tile inside the traced area (they come first)
{
#if PT_TILE_8X8
    return kWave;
#else
    return min(kWave, fp.traced_width - (tile % fp.x_blocks) * kWave);
#endif
}

template <int SLOTS>
struct TriRegs { float p1x[SLOTS], p1y[SLOTS], p1z[SLOTS], e1x[SLOTS], e1y[SLOTS], e1z[SLOTS], e2x[SLOTS], e2y[SLOTS], e2z[SLOTS]; };

template <int SLOTS>
__device__ __forceinline__ void load_chunk(TriRegs<SLOTS> &T, const float *__restrict__ soa, int chunk, int lane)
{
    const float *b = soa + (size_t)chunk * 9 * (kWave * SLOTS) + lane;
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        T.p1x[s] = b[0 * kWave * SLOTS + s * kWave]; T.p1y[s] = b[1 * kWave * SLOTS + s * kWave]; T.p1z[s] = b[2 * kWave * SLOTS + s * kWave];
        T.e1x[s] = b[3 * kWave * SLOTS + s * kWave]; T.e1y[s] = b[4 * kWave * SLOTS + s * kWave]; T.e1z[s] = b[5 * kWave * SLOTS + s * kWave];
        T.e2x[s] = b[6 * kWave * SLOTS + s * kWave]; T.e2y[s] = b[7 * kWave * SLOTS + s * kWave]; T.e2z[s] = b[8 * kWave * SLOTS + s * kWave];
    }
}

// Load through the constant address space: the scene arrays are never written while a kernel runs, and
// with a wave-uniform index hipcc then issues an s_load (scalar cache, result in SGPRs) instead of 64
// identical vector loads.
__device__ __forceinline__ float4 uniform_load(const float4 *p, int i)
{
#if PT_SLOAD
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(4))) const f4v cf4v;
    const f4v v = *reinterpret_cast<cf4v *>(reinterpret_cast<uintptr_t>(p + i));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return p[i];
#endif
}

// Wave votes straight from the condition: HIP's __ballot / __any take an int, and the bool -> int -> "!= 0" round trip
// survives optimisation in places as a v_cndmask + v_cmp pair (8 issue cycles) per vote -- two of them per traversal step.
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// Copy a wave-uniform value into a VGPR and hide its uniformity from the compiler.  On gfx950 a VALU instruction with
// an SGPR source issues in 4 SIMD cycles, the same mul/add/sub/fma with VGPR sources in 2 (tools/valu_mix.hip), so the
// broadcast ray lives in VGPRs: 6 moves per ray instead of 12 slow operand reads per triangle slot.
// Optional per-phase cycle accounting (s_memtime per wave, summed into counters 12..15): diagnostic builds only.
#ifndef PT_PHASE_TIMERS
#define PT_PHASE_TIMERS 0
#endif
#if PT_PHASE_TIMERS == 1
#define PT_TICK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PT_ACC(acc, a, b) acc += (b) - (a)
#define PT_COUNT(acc, n)
#elif PT_PHASE_TIMERS == 2     // event counts instead of cycles: [0] traversal phases [1] node steps [2] triangle steps [3] rays at phase entry
#define PT_TICK(var)
#define PT_ACC(acc, a, b)
#define PT_COUNT(acc, n) acc += (n)
#else
#define PT_COUNT(acc, n)
#define PT_TICK(var)
#define PT_ACC(acc, a, b)
#endif
#ifndef PT_DUMMY_KIND
#define PT_DUMMY_KIND 0
#endif
#ifndef PT_RAY_IN_VGPR
#define PT_RAY_IN_VGPR 1
#endif
__device__ __forceinline__ float to_vgpr(float s)
{
#if PT_RAY_IN_VGPR
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
#else
    return s;
#endif
}

// wave64 minimum through DPP (no LDS): prefix-min inside each row of 16, then row broadcasts; the
// result is valid in lane 63.  min is idempotent, so full row/bank masks with old = self are safe.
__device__ __forceinline__ float wave_min_to_lane63(float v)
{
#define PT_DPP_MIN(ctrl, rmask)                                                                                         \
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false)))
    PT_DPP_MIN(0x111, 0xf);  // row_shr:1
    PT_DPP_MIN(0x112, 0xf);  // row_shr:2
    PT_DPP_MIN(0x114, 0xf);  // row_shr:4
    PT_DPP_MIN(0x118, 0xf);  // row_shr:8
    PT_DPP_MIN(0x142, 0xa);  // row_bcast:15 -> rows 1,3
    PT_DPP_MIN(0x143, 0xc);  // row_bcast:31 -> rows 2,3
#undef PT_DPP_MIN
    return v;
}

// Where a lane's SLOTS mesh triangles live while rays are tested against them.
//  RegStore: in VGPRs (any mesh size; a mesh of more than one chunk is re-loaded chunk by chunk per ray batch).
//  LdsStore: in a workgroup-shared LDS image of chunk 0 (meshes of one chunk), read back slot by slot with two
//            ds_read_b128 + one ds_read_b32 per triangle.  Frees ~100 VGPRs -> more waves per SIMD, which is what
//            the loop needs: one wave issues a VALU op only every ~5 cycles (tools/loop_bench.hip, tools/valu_mix.hip).
template <int SLOTS>
struct RegStore {
    static constexpr bool is_bvh = false;
    TriRegs<SLOTS> T;
    const float *soa; int lane; bool resident;
    __device__ __forceinline__ void init(const SceneView &sc, int lane_)
    {
        soa = sc.mesh_soa; lane = lane_; resident = sc.n_chunks == 1;
        if (resident) load_chunk<SLOTS>(T, soa, 0, lane);
    }
    __device__ __forceinline__ void begin_chunk(int chunk) { if (!resident) load_chunk<SLOTS>(T, soa, chunk, lane); }
    __device__ __forceinline__ void get(int s, v3 &p1, v3 &e1, v3 &e2) const
    {
        p1 = V(T.p1x[s], T.p1y[s], T.p1z[s]); e1 = V(T.e1x[s], T.e1y[s], T.e1z[s]); e2 = V(T.e2x[s], T.e2y[s], T.e2z[s]);
    }
};

template <int SLOTS>
struct LdsStore {
    static constexpr bool is_bvh = false;
    const float4 *A, *B; const float *C;      // LDS images, already offset by this lane: [slot][64]
    __device__ __forceinline__ void begin_chunk(int) {}
    __device__ __forceinline__ void get(int s, v3 &p1, v3 &e1, v3 &e2) const
    {
        const float4 a = A[s * kWave], b = B[s * kWave];
        const float c = C[s * kWave];
        p1 = V(a.x, a.y, a.z); e1 = V(a.w, b.x, b.y); e2 = V(b.z, b.w, c);
    }
};

// One broadcast ray against the SLOTS triangles this lane holds: local closest hit (lt, ls).
template <int SLOTS, bool FAST, class Store>
__device__ __forceinline__ void test_slots(const Store &st, v3 ro, v3 rd, float &lt, int &ls, bool &bad, float hl)
{
    lt = kInfinity;
    ls = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        v3 p1, e1, e2;
        st.get(s, p1, e1, e2);
        float t, det;
        bool has = hit_triangle_t<FAST>(p1, e1, e2, ro, rd, t, det, bad, hl);
        if (has && t < lt) { lt = t; ls = s; }
        // keep the slots (nearly) sequential: interleaving 11 independent tests costs ~20 live VGPRs each
        if ((s + 1) % PT_SLOT_GROUP == 0) __builtin_amdgcn_sched_barrier(PT_SCHED_MASK);
    }
}

// Closest mesh hit for every lane whose bit is set in `mask` (wave-uniform).  best_t/best_id are the
// per-pixel-lane results (kInfinity / -1 when nothing is hit).  rt.comp:213-218.
template <int SLOTS, class Store>
__device__ __forceinline__ void mesh_closest(const SceneView &sc, Store &st, int lane, unsigned long long mask,
                                             v3 o, v3 d, float &best_t, int &best_id)
{
    for (int chunk = 0; chunk < sc.n_chunks; chunk++) {
        st.begin_chunk(chunk);
        unsigned long long m = mask;
        while (m) {
            const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
            m &= m - 1;
            // broadcast the ray through SGPRs
            const v3 ro = V(to_vgpr(readlane_f(o.x, src)), to_vgpr(readlane_f(o.y, src)), to_vgpr(readlane_f(o.z, src)));
            const v3 rd = V(to_vgpr(readlane_f(d.x, src)), to_vgpr(readlane_f(d.y, src)), to_vgpr(readlane_f(d.z, src)));
            float lt;
            int ls;
            bool bad = false;
#if PT_FAST_RCP
            test_slots<SLOTS, true>(st, ro, rd, lt, ls, bad, sc.sh.hit_limit);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) test_slots<SLOTS, false>(st, ro, rd, lt, ls, bad, sc.sh.hit_limit);
#else
            test_slots<SLOTS, false>(st, ro, rd, lt, ls, bad, sc.sh.hit_limit);
#endif
            const float tmin = readlane_f(wave_min_to_lane63(lt), 63);
            if (tmin < kInfinity) {
                const unsigned long long winners = wave_ballot(lt == tmin);
                const int wl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(winners));
                const int slot = __builtin_amdgcn_readlane(ls, wl);
                const int id = chunk * (kWave * SLOTS) + wl * SLOTS + slot;
                const float cur = readlane_f(best_t, src);
                if (tmin < cur && lane == src) { best_t = tmin; best_id = id; }
            }
        }
    }
}

// ---- BVH traversal (one ray per lane) ----------------------------------------------------------------------
// Finds the same (t, id) as the brute-force scan: every triangle whose test can pass is visited, the test itself is
// the reference arithmetic (hit_triangle_t), and the winner is the lexicographic minimum of (t, original index).
// Only the SKIPPING is ours, so its arithmetic is free to use FMAs; it must merely be conservative:
//  * boxes are grown per ray by delta = 4e-6 * (|o|_1 + mesh scale) (~30 ulp of the largest coordinate in play): the
//    reference's float test can accept a ray that geometrically misses a triangle by rounding noise of that size;
//  * a subtree is pruned against the best hit only if its entry distance exceeds best_t * 1.001 + delta, since the
//    reference's computed t carries a relative error that grows with 1/sin(angle between ray and triangle);
//  * min/max are IEEE minNum/maxNum, so a 0 * inf = NaN slab (ray parallel to an axis and lying on the slab plane)
//    imposes no constraint.
// Not covered by construction: a ray within rounding noise of being COPLANAR with a triangle it misses in-plane (the
// reference's u, v are then noise / noise); see DESIGN.md.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f4v lds_cf4v;

// BVH in global memory (any size; served by L1/L2) ...
struct BvhStore {
    static constexpr bool is_bvh = true;
    unsigned *stack;          // LDS, already offset by this lane; entry e at stack[e * 64]
    const float4 *nodes, *tris;
    __device__ __forceinline__ void begin_chunk(int) {}
    __device__ __forceinline__ void node(int i, float4 &a, float4 &b, float4 &c, int &kl, int &kr) const
    {
        const float4 *n = nodes + 4 * (size_t)i;
        a = n[0]; b = n[1]; c = n[2];
        const float4 k = n[3];
        kl = __float_as_int(k.x); kr = __float_as_int(k.y);
    }
    // four-wide node in its 64-byte quantised form (pt_bvh.hpp BvhNode4Q): four loads per lane, then every plane is rebuilt
    // as fmaf(q, 2^e, o) (the byte extraction + conversion compiles to one v_cvt_f32_ubyteN) -- the operation the host checked its outward rounding with.  Child k is component k of the six
    // box vectors.  An unused slot has no box of its own here: the caller tests its code (kChildTest).
    static constexpr bool kChildTest = true;
    const float4 *nodes_q;
    __device__ __forceinline__ void node4(int i, float4 &mnx, float4 &mny, float4 &mnz, float4 &mxx, float4 &mxy, float4 &mxz, int &c0, int &c1, int &c2, int &c3) const
    {
        const float4 *n = nodes_q + 4 * (size_t)i;
        const float4 o = n[0], k = n[1], qa = n[2];
        const float2 qb = *reinterpret_cast<const float2 *>(n + 3);
        c0 = __float_as_int(k.x); c1 = __float_as_int(k.y); c2 = __float_as_int(k.z); c3 = __float_as_int(k.w);
        const unsigned ex = __float_as_uint(o.w);
        const float sx = __uint_as_float((ex & 0xffu) << 23), sy = __uint_as_float((ex & 0xff00u) << 15), sz = __uint_as_float((ex & 0xff0000u) << 7);
#define PT_DEQ(word, step, org) make_float4(fmaf((float)(((word) >> 0) & 0xffu), step, org), fmaf((float)(((word) >> 8) & 0xffu), step, org), \
                                            fmaf((float)(((word) >> 16) & 0xffu), step, org), fmaf((float)(((word) >> 24) & 0xffu), step, org))
        mnx = PT_DEQ(__float_as_uint(qa.x), sx, o.x); mny = PT_DEQ(__float_as_uint(qa.y), sy, o.y); mnz = PT_DEQ(__float_as_uint(qa.z), sz, o.z);
        mxx = PT_DEQ(__float_as_uint(qa.w), sx, o.x); mxy = PT_DEQ(__float_as_uint(qb.x), sy, o.y); mxz = PT_DEQ(__float_as_uint(qb.y), sz, o.z);
#undef PT_DEQ
    }
    // leaf-ordered triangle records, 64 B each (one cache line): (p1 | canonical index), (e1 | material), (e2 | 0), (normal | 0).
    // The traversal reads the first three; the resolve phase reads the record of the winning triangle again -- the line the
    // leaf step has just brought in -- instead of a fourth, cold line of the canonical table (kLeafRecords)
    static constexpr bool kLeafRecords = true;
    // the same node with its planes sorted by the ray's direction (sx, sy, sz = 1: the ray runs towards smaller coordinates and
    // enters every box through its max plane): near planes, far planes -- the slab test then needs no min / max per axis
    // The distance to a plane is then t = plane * m + a with the ray's (m, a) = (inv, nn | nf).  Here the plane is q * 2^e + o, so
    // the dequantisation folds into the ray's constants per node: t = q * (2^e * inv) + (o * inv + a) -- the scaling by a power of
    // two is exact, the sum is rounded once more than the host's check of the outward rounding assumed (an ulp of |o * inv|,
    // far inside the traversal's margin delta * inv) -- 24 FMAs per visit instead of 48.
    __device__ __forceinline__ void node4_dir(int i, unsigned sx, unsigned sy, unsigned sz, float4 &nx, float4 &ny, float4 &nz, float4 &fx, float4 &fy, float4 &fz,
                                              int &c0, int &c1, int &c2, int &c3, v3 &m, v3 &an, v3 &af) const
    {
        const float4 *n = nodes_q + 4 * (size_t)i;
        const float4 o = n[0], k = n[1], qa = n[2];
        const float2 qb = *reinterpret_cast<const float2 *>(n + 3);
        c0 = __float_as_int(k.x); c1 = __float_as_int(k.y); c2 = __float_as_int(k.z); c3 = __float_as_int(k.w);
        const unsigned ex = __float_as_uint(o.w);
        const float stx = __uint_as_float((ex & 0xffu) << 23), sty = __uint_as_float((ex & 0xff00u) << 15), stz = __uint_as_float((ex & 0xff0000u) << 7);
        const unsigned lx = __float_as_uint(qa.x), ly = __float_as_uint(qa.y), lz = __float_as_uint(qa.z);
        const unsigned hx = __float_as_uint(qa.w), hy = __float_as_uint(qb.x), hz = __float_as_uint(qb.y);
#define PT_DEQ(word, step, org) make_float4(fmaf((float)(((word) >> 0) & 0xffu), step, org), fmaf((float)(((word) >> 8) & 0xffu), step, org), \
                                            fmaf((float)(((word) >> 16) & 0xffu), step, org), fmaf((float)(((word) >> 24) & 0xffu), step, org))
#undef PT_DEQ
#define PT_BYTES(word) make_float4((float)(((word) >> 0) & 0xffu), (float)(((word) >> 8) & 0xffu), (float)(((word) >> 16) & 0xffu), (float)(((word) >> 24) & 0xffu))
        nx = PT_BYTES(sx ? hx : lx); ny = PT_BYTES(sy ? hy : ly); nz = PT_BYTES(sz ? hz : lz);
        fx = PT_BYTES(sx ? lx : hx); fy = PT_BYTES(sy ? ly : hy); fz = PT_BYTES(sz ? lz : hz);
#undef PT_BYTES
        an = V(fmaf(o.x, m.x, an.x), fmaf(o.y, m.y, an.y), fmaf(o.z, m.z, an.z));
        af = V(fmaf(o.x, m.x, af.x), fmaf(o.y, m.y, af.y), fmaf(o.z, m.z, af.z));
        m = V(m.x * stx, m.y * sty, m.z * stz);
    }
    __device__ __forceinline__ void tri(int j, float4 &p1, float4 &e1, float4 &e2) const
    {
        const float4 *t = tris + kBvhTriVecs * (size_t)j;
        p1 = t[0]; e1 = t[1]; e2 = t[2];
    }
    const float4 *sph;
    __device__ __forceinline__ void sphere(int j, float4 &cr, int &index) const { cr = sph[2 * j]; index = __float_as_int(sph[2 * j + 1].x); }
    lds_cf4v *scene = nullptr;      // wave-uniform operands staged in LDS (stage_scene_lds), null: read through scalar loads
    lds_cf4v *shade = nullptr;      // the kShadeLdsVecs shading vectors in front of them (always staged)
};
// ... or copied into the workgroup's LDS (meshes whose nodes + triangles fit beside the stacks): a dependent node
// fetch then costs ~100 cycles instead of an L2 round trip, and the traversal is a chain of them.
struct BvhLdsStore {
    static constexpr bool is_bvh = true;
    unsigned *stack;
    lds_cf4v *nodes, *tris;
    __device__ __forceinline__ void begin_chunk(int) {}
    static __device__ __forceinline__ float4 f4(const f4v v) { return make_float4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ void node(int i, float4 &a, float4 &b, float4 &c, int &kl, int &kr) const
    {
        lds_cf4v *n = nodes + 4 * i;
        a = f4(n[0]); b = f4(n[1]); c = f4(n[2]);
        const f4v k = n[3];
        kl = __float_as_int(k.x); kr = __float_as_int(k.y);
    }
    // four-wide node, float boxes (pt_bvh.hpp BvhNode4, 128 B): child k is component k of the six box vectors; an unused slot
    // carries a box no ray meets
    static constexpr bool kChildTest = false;
    __device__ __forceinline__ void node4(int i, float4 &mnx, float4 &mny, float4 &mnz, float4 &mxx, float4 &mxy, float4 &mxz, int &c0, int &c1, int &c2, int &c3) const
    {
        lds_cf4v *n = nodes + kBvhLdsNodeVecs * i;
        mnx = f4(n[0]); mny = f4(n[1]); mnz = f4(n[2]); mxx = f4(n[3]); mxy = f4(n[4]); mxz = f4(n[5]);
        const f4v k = n[6];
        c0 = __float_as_int(k.x); c1 = __float_as_int(k.y); c2 = __float_as_int(k.z); c3 = __float_as_int(k.w);
    }
    // planes sorted by the ray's direction (BvhStore::node4_dir): a per-lane choice of which of the six box vectors to read
    __device__ __forceinline__ void node4_dir(int i, unsigned sx, unsigned sy, unsigned sz, float4 &nx, float4 &ny, float4 &nz, float4 &fx, float4 &fy, float4 &fz,
                                              int &c0, int &c1, int &c2, int &c3, v3 &, v3 &, v3 &) const      // float planes: the ray's (m, a) as they are
    {
        lds_cf4v *n = nodes + kBvhLdsNodeVecs * i;
        nx = f4(n[3 * sx]); ny = f4(n[1 + 3 * sy]); nz = f4(n[2 + 3 * sz]);
        fx = f4(n[3 - 3 * sx]); fy = f4(n[4 - 3 * sy]); fz = f4(n[5 - 3 * sz]);
        const f4v k = n[6];
        c0 = __float_as_int(k.x); c1 = __float_as_int(k.y); c2 = __float_as_int(k.z); c3 = __float_as_int(k.w);
    }
    // the LDS copy keeps the first three vectors of each record; the resolve phase goes to the canonical table
    static constexpr bool kLeafRecords = false;
    __device__ __forceinline__ void tri(int j, float4 &p1, float4 &e1, float4 &e2) const
    {
        lds_cf4v *t = tris + 3 * j;
        p1 = f4(t[0]); e1 = f4(t[1]); e2 = f4(t[2]);
    }
    lds_cf4v *sph;
    __device__ __forceinline__ void sphere(int j, float4 &cr, int &index) const { cr = f4(sph[2 * j]); index = __float_as_int(sph[2 * j + 1].x); }
    lds_cf4v *scene = nullptr;      // wave-uniform operands staged in LDS (stage_scene_lds), null: read through scalar loads
    lds_cf4v *shade = nullptr;      // the kShadeLdsVecs shading vectors in front of them (always staged)
};

__device__ __forceinline__ void slab_test(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, v3 olo, v3 ohi, v3 inv,
                                          float &t0, float &t1)
{
    // ((b -+ delta) - o) * inv for both planes of each axis, with olo = o + delta, ohi = o - delta; conservative, not
    // reference arithmetic.  0 * inf = NaN (ray parallel to the axis, origin on the grown plane) drops out of minNum/maxNum.
    const float x1 = (mnx - olo.x) * inv.x, x2 = (mxx - ohi.x) * inv.x;
    const float y1 = (mny - olo.y) * inv.y, y2 = (mxy - ohi.y) * inv.y;
    const float z1 = (mnz - olo.z) * inv.z, z2 = (mxz - ohi.z) * inv.z;
    t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2)), __builtin_fmaxf(__builtin_fminf(z1, z2), 0.0f));
    t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2)), __builtin_fmaxf(z1, z2)) * 1.000001f;
}

// The same test with the origin folded in: t = b * inv + (-(o +- delta) * inv), one FMA per plane, three-input min/max.
// A ray that starts inside the grown box has t0 < 0 <= t1.  The folded form must never see an infinite `inv`: b * inf and
// -(o * inf) are infinities of opposite sign for a box that straddles the origin's coordinate, and inf - inf = NaN on ONE
// plane only would leave the other plane's infinity as a false "miss".  The caller therefore passes slab_inv(d), which is
// 1/d limited to +-1e19: with |coordinates| < 1e18 (checked at upload) no product overflows, and shrinking |1/d| only
// moves the slab's entry and exit towards t = 0 in proportion -- a ray that crawls along an axis at less than 1e-19 per
// unit t cannot leave or enter the grown slab before t = 1e7 (kInfinity) unless it starts within 1e-12 of its surface,
// and the box proper lies a full delta further in.
__device__ __forceinline__ float slab_inv(float d) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d), -1e19f, 1e19f); }
__device__ __forceinline__ void slab_test_fma(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, v3 nlo, v3 nhi, v3 inv,
                                              float &t0, float &t1)
{
    const float x1 = fmaf(mnx, inv.x, nlo.x), x2 = fmaf(mxx, inv.x, nhi.x);
    const float y1 = fmaf(mny, inv.y, nlo.y), y2 = fmaf(mxy, inv.y, nhi.y);
    const float z1 = fmaf(mnz, inv.z, nlo.z), z2 = fmaf(mxz, inv.z, nhi.z);
    t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2)), __builtin_fminf(z1, z2));
    t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2)), __builtin_fmaxf(z1, z2)) * 1.000001f;
}

template <class Store>
__device__ __forceinline__ void bvh_closest(const SceneView &sc, Store &st, bool gate, v3 o, v3 d, float &best_t, int &best_id,
                                            unsigned &n_nodes, unsigned &n_tests)
{
    if (!gate) return;
    const v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float delta = 4e-6f * ((fabsf(o.x) + fabsf(o.y)) + (fabsf(o.z) + sc.bvh_scale));
    const v3 olo = V(o.x + delta, o.y + delta, o.z + delta), ohi = V(o.x - delta, o.y - delta, o.z - delta);
    unsigned *stk = st.stack;
    int sp = 0, cur = 0;
    for (;;) {
        int next;
        n_nodes++;
        const float lim = fmaf(best_t, 1.001f, delta);
        if constexpr (kBvhWide) {
            // four children, visited in slot order (this routine serves the test-only primary-hit probe; the trace kernels'
            // traversal phase goes to the nearest child first)
            float4 mnx, mny, mnz, mxx, mxy, mxz;
            int cc[4];
            st.node4(cur, mnx, mny, mnz, mxx, mxy, mxz, cc[0], cc[1], cc[2], cc[3]);
            const float bx[4][6] = {{mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x}, {mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y},
                                    {mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z}, {mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w}};
            bool have = false;
            next = 0;
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                float t0, t1;
                slab_test(bx[k][0], bx[k][1], bx[k][2], bx[k][3], bx[k][4], bx[k][5], olo, ohi, inv, t0, t1);
                if (t0 <= t1 && t0 <= lim && (!Store::kChildTest || cc[k] != kTravDone)) {
                    if (have) { stk[sp * kWave] = (unsigned)next; sp++; }
                    next = cc[k]; have = true;
                }
            }
            if (!have) {
                if (sp == 0) break;
                sp--;
                next = (int)stk[sp * kWave];
            }
        } else {
        float4 a, b, c;
        int kl, kr;
        st.node(cur, a, b, c, kl, kr);
        float l0, l1, r0, r1;
        slab_test(a.x, a.y, a.z, a.w, b.x, b.y, olo, ohi, inv, l0, l1);
        slab_test(b.z, b.w, c.x, c.y, c.z, c.w, olo, ohi, inv, r0, r1);
        const bool hl = l0 <= l1 && l0 <= lim, hr = r0 <= r1 && r0 <= lim;
        if (hl && hr) {
            const bool left_first = l0 <= r0;
            next = left_first ? kl : kr;
            stk[sp * kWave] = (unsigned)(left_first ? kr : kl);
            sp++;
        } else if (hl) {
            next = kl;
        } else if (hr) {
            next = kr;
        } else {
            if (sp == 0) break;
            sp--;
            next = (int)stk[sp * kWave];
        }
        }
        bool done = false;
        while (next < 0) {
            const int code = ~next, first = code >> 3, cnt = (code & 7) + 1;
            for (int i = 0; i < cnt; i++) {
                float4 p1, e1, e2;
                st.tri(first + i, p1, e1, e2);
                float t, det;
                bool bad = false;
                bool has = hit_triangle_t<true>(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, bad, sc.sh.hit_limit);
                if (__builtin_expect(bad, 0)) has = hit_triangle_ieee(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, sc.sh.hit_limit);
                const int id = __float_as_int(p1.w);
                n_tests++;
                if (has && (t < best_t || (t == best_t && id < best_id))) { best_t = t; best_id = id; }
            }
            if (sp == 0) { done = true; break; }
            sp--;
            next = (int)stk[sp * kWave];
        }
        if (done) break;
        cur = next;
    }
}

struct Hit { int id; float t; bool front; v3 normal; uint32_t mat; };

// Scene triangles (SSBO 5) then spheres (SSBO 4), in order, strict `t < closest` (rt.comp:219-230), starting from
// the closest hit found so far (t, id).
__device__ __forceinline__ void scene_closest(const SceneView &sc, v3 o, v3 d, float &best_t, int &best_id, bool with_spheres = true)
{
    for (int i = 0; i < sc.n_tris; i++) {
        const float4 p1 = uniform_load(sc.tri_p1, i), e1 = uniform_load(sc.tri_e1, i), e2 = uniform_load(sc.tri_e2, i);
        float t, det;
        bool has = hit_triangle(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, sc.sh.hit_limit);
        if (has && t < best_t) { best_t = t; best_id = sc.n_mesh + i; }
    }
    if (!with_spheres) return;
    for (int i = 0; i < sc.n_sph; i++) {
        const float4 s = uniform_load(sc.sph, i);
        float t; bool front;
        bool has = hit_sphere(V(s.x, s.y, s.z), s.w, o, d, t, front, sc.sh.hit_limit, (sc.sh.flags & kShSumXyz) != 0);
        if (has && t < best_t) { best_t = t; best_id = sc.n_mesh + sc.n_tris + i; }
    }
}

// HitInfo of the winner (rt.comp:144-147, 171-177): front face, normal, material.  det / front are recomputed with the
// operations the test itself used, so they are the values the scan saw.
__device__ __forceinline__ void finish_hit(const SceneView &sc, v3 o, v3 d, Hit &h)
{
    if (h.id < 0) return;
    if (h.id >= sc.n_mesh + sc.n_tris) {
        const int i = h.id - sc.n_mesh - sc.n_tris;
        const float4 s = sc.sph[i];
        float t; bool front;
        hit_sphere(V(s.x, s.y, s.z), s.w, o, d, t, front, sc.sh.hit_limit, (sc.sh.flags & kShSumXyz) != 0);
        v3 point = o + d * h.t;                                         // rt.comp:144
        v3 n = normalize(point - V(s.x, s.y, s.z));
        h.front = front;
        h.normal = front ? n : -n;                                      // rt.comp:147
        h.mat = sc.sph_mat[i];
    } else {
        float4 p1, e1, e2, nr;
        if (h.id >= sc.n_mesh) {
            const int i = h.id - sc.n_mesh;
            p1 = sc.tri_p1[i]; e1 = sc.tri_e1[i]; e2 = sc.tri_e2[i]; nr = sc.tri_nrm[i];
        } else {
            p1 = sc.mesh_p1[h.id]; e1 = sc.mesh_e1[h.id]; e2 = sc.mesh_e2[h.id]; nr = sc.mesh_nrm[h.id];
        }
        float t, det;
        hit_triangle(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, sc.sh.hit_limit);
        v3 n = V(nr.x, nr.y, nr.z);
        h.front = !(det < 0.0f);                                        // rt.comp:174
        h.normal = det < 0.0f ? n : -n;                                 // rt.comp:177
        h.mat = __float_as_uint(p1.w);
    }
}

// The same two steps for the traversal kernels, from the unified table (one base pointer, one cache line per record).
// xyz: the sphere test's summation order (kShSumXyz), passed in so that a kernel built for rt.comp's forms carries no select
__device__ __forceinline__ void scene_closest_tab(const SceneView &sc, v3 o, v3 d, float &best_t, int &best_id, bool with_spheres, bool xyz)
{
    for (int i = 0; i < sc.n_tris; i++) {
        const int rec = 4 * (sc.n_mesh + i);
        const float4 p1 = uniform_load(sc.prim_tab, rec), e1 = uniform_load(sc.prim_tab, rec + 1), e2 = uniform_load(sc.prim_tab, rec + 2);
        float t, det;
        bool has = hit_triangle(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, sc.sh.hit_limit);
        if (has && t < best_t) { best_t = t; best_id = sc.n_mesh + i; }
    }
    if (!with_spheres) return;
    for (int i = 0; i < sc.n_sph; i++) {
        const float4 s = uniform_load(sc.sph, i);
        float t; bool front;
        bool has = hit_sphere(V(s.x, s.y, s.z), s.w, o, d, t, front, sc.sh.hit_limit, xyz);
        if (has && t < best_t) { best_t = t; best_id = sc.n_mesh + sc.n_tris + i; }
    }
}
// leaf_rec >= 0: the hit is mesh triangle h.id and its leaf-ordered record (BvhStore::kLeafRecords) is sc.bvh_tris[leaf_rec]
__device__ __forceinline__ void finish_hit_tab(const SceneView &sc, v3 o, v3 d, Hit &h, bool xyz, int leaf_rec = -1)
{
    if (h.id < 0) return;
    if (h.id >= sc.n_mesh + sc.n_tris) {
        const int i = h.id - sc.n_mesh - sc.n_tris;
        const float4 s = sc.sph[i];
        float t; bool front;
        hit_sphere(V(s.x, s.y, s.z), s.w, o, d, t, front, sc.sh.hit_limit, xyz);
        v3 point = o + d * h.t;                                         // rt.comp:144
        v3 n = normalize(point - V(s.x, s.y, s.z));
        h.front = front;
        h.normal = front ? n : -n;                                      // rt.comp:147
        h.mat = sc.sph_mat[i];
    } else {
        const bool leaf = leaf_rec >= 0;
        const float4 *r = leaf ? sc.bvh_tris + kBvhTriVecs * (size_t)leaf_rec : sc.prim_tab + 4 * (size_t)h.id;
        const float4 p1 = r[0], e1 = r[1], e2 = r[2], nr = r[3];
        float t, det;
        hit_triangle(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, sc.sh.hit_limit);
        v3 n = V(nr.x, nr.y, nr.z);
        h.front = !(det < 0.0f);                                        // rt.comp:174
        h.normal = det < 0.0f ? n : -n;                                 // rt.comp:177
        h.mat = __float_as_uint(leaf ? e1.w : p1.w);
    }
}

// CalculateRay (rt.comp:204-232) for the pixel of this lane; `alive` lanes take part, the mesh part is
// wave-cooperative.  Canonical ids: mesh i | n_mesh + i | n_mesh + n_tris + i | -1.
template <int SLOTS, class Store>
__device__ __forceinline__ Hit calculate_ray(const SceneView &sc, Store &st, int lane, bool alive, bool primary, v3 o, v3 d,
                                             unsigned &n_seg, unsigned &n_mesh_seg, unsigned &n_prim_mesh_seg,
                                             unsigned &n_bvh_nodes, unsigned &n_bvh_tests, unsigned long long &cyc_mesh)
{
    Hit h;
    h.id = -1; h.t = kInfinity; h.front = false; h.normal = V(0, 0, 0); h.mat = 0;
    bool gate = false;
    if (alive && sc.n_mesh > 0 && (sc.sh.flags & kShTraceMesh))
        gate = hit_aabb(V(sc.bmin[0], sc.bmin[1], sc.bmin[2]), V(sc.bmax[0], sc.bmax[1], sc.bmax[2]), o, d);
    const unsigned long long gate_mask = wave_ballot(gate);
    n_seg += __popcll(wave_ballot(alive));
    n_mesh_seg += __popcll(gate_mask);
    n_prim_mesh_seg += __popcll(wave_ballot(gate && primary));
    if constexpr (Store::is_bvh) {
        PT_TICK(b0);
        bvh_closest(sc, st, gate, o, d, h.t, h.id, n_bvh_nodes, n_bvh_tests);
        PT_TICK(b1);
        PT_ACC(cyc_mesh, b0, b1);
    } else {
        if (gate_mask) mesh_closest<SLOTS>(sc, st, lane, gate_mask, o, d, h.t, h.id);
    }
    if (!alive) return h;

    scene_closest(sc, o, d, h.t, h.id);
    finish_hit(sc, o, d, h);
    return h;
}

// rt.comp:326-333.  Returns the rng state after the six jitter draws.
__device__ __forceinline__ uint32_t primary_ray(const FrameParams &fp, const Shading &sh, int x, int y, int pass, v3 &o, v3 &d)
{
    const float W = (float)fp.width, H = (float)fp.height;
    const float fx = (float)x + sh.pix_off, fy = (float)y + sh.pix_off;       // rt.comp:12 | rt.frag:14 (x + 0.5 is exact)
    const float vx = fx / W, vy = fy / H;
    const float px = ((vx - 0.5f) * W) / H;            // rt.comp:14,17
    const float py = vy - 0.5f;
    uint32_t state = (uint32_t)pass * pixel_index(fx, fy, W, H);        // rt.comp:327
    // `RandomDirection(s) * 0.00025` (rt.comp:332): the reference's compiler folds the constant into the
    // normalize as (v * 0.00025) * (1/sqrt(dot(v,v)))
    float gx = random_value_nd(state), gy = random_value_nd(state), gz = random_value_nd(state);
    v3 g = V(gx, gy, gz);
    float gr = 1.0f / sqrtf(dot(g, g));
    v3 jit = (g * sh.jitter_scale) * gr;
    const float X = -px + jit.x, Y = py + jit.y;
    const float *m = fp.iv;   // c0 = m[0..2], c1 = m[3..5], c2 = m[6..8]
    // mat4 * vec4(X, Y, 1, 0) -> (c0*X + c2) + c1*Y
    d = normalize(V((m[0] * X + m[6]) + m[3] * Y, (m[1] * X + m[7]) + m[4] * Y, (m[2] * X + m[8]) + m[5] * Y));
    o = V(fp.cam[0], fp.cam[1], fp.cam[2]);
    return state;
}

// One bounce of RayTrace's loop body for a path that hit (rt.comp:276-311).
struct PathState { v3 o, d, ray_color, incoming, emitted; uint32_t rng; int bounces; };

__device__ __forceinline__ void shade_hit(const SceneView &sc, const Hit &h, PathState &ps)
{
    ps.bounces++;
    const PtMaterial *m = &sc.mats[h.mat];
    const float m_ior = m->ior, m_smooth = m->smoothness, m_spec_smooth = m->specularSmoothness;
    v3 o = ps.o + ps.d * h.t;                                           // hitPoint (rt.comp:175 / :144)
    const float ior = h.front ? (1.0f / m_ior) : m_ior;
    const bool is_spec = m->specularChance > random_value(ps.rng);
    const float fres = fresnel_reflect_amount((sc.sh.flags & kShN1One) ? 1.0f : ior, m_ior, h.normal, ps.d, 1.0f - m->refractionAmount);   // rt.comp:283 | rt.frag:283
    const bool is_refr = fres < random_value(ps.rng);
    const v3 hemi = random_in_hemisphere(h.normal, ps.rng);
    const v3 hemi2 = (sc.sh.flags & kShTwoDraws) ? random_in_hemisphere(h.normal, ps.rng) : hemi;     // rt.frag:295
    const v3 a = mix3(hemi, reflect3(ps.d, h.normal), is_spec ? m_spec_smooth : m_smooth);
    const v3 b = mix3(-hemi2, refract3(ps.d, h.normal, ior), m_smooth);
    ps.d = normalize(select3(is_refr, b, a));
    ps.o = o + ps.d * sc.sh.origin_offset;                              // rt.comp:305
    const v3 base = V(m->baseColor[0], m->baseColor[1], m->baseColor[2]);
    const v3 spec = V(m->specularColor[0], m->specularColor[1], m->specularColor[2]);
    ps.emitted = ps.emitted + V(m->emissionColor[0], m->emissionColor[1], m->emissionColor[2]) * m->emissionStrength;
    ps.ray_color = ps.ray_color * select3(is_refr, base, select3(is_spec, spec, base));
    ps.incoming = ps.incoming + ps.emitted * ps.ray_color;
}

// ------------------------------------------------------------------------------------------------
// The hot kernel.  Persistent single-wave workgroups (grid = what the chip holds at 3 waves/SIMD) pull
// work items from an atomic queue; the mesh triangles are loaded into registers once per wave for
// the whole launch.  Work item = one tile of 64 pixels (tile_pixel) x `fp.chunk` consecutive
// passes.  Each lane owns one pixel and walks through its passes on its own ("persistent lane": when
// a path ends the lane starts the primary ray of its next pass at once), so lanes stay busy although
// path lengths differ; a sample is a pure function of (pixel, pass) (rt.comp:326-327), so this changes
// no result.  The 1-spp radiance of every (pass, pixel) goes to fp.radiance; pt_accumulate applies the
// reference's running mean in pass order (rt.comp:337-339).  Items are ordered tile-major over the tiles
// sorted heaviest first (pt_tile_weights / pt_sort_tiles), so the launch ends in one-iteration sky tiles and the
// persistent waves finish together.
template <int SLOTS, class Store>
__device__ __forceinline__ void trace_items_body(const SceneView &sc, const FrameParams &fp, Store &st, int lane)
{
    const int max_bounces = fp.debug_normal ? 1 : fp.max_bounces;      // rt.comp:23
    const size_t plane = (size_t)fp.width * fp.height;
    unsigned n_seg = 0, n_mesh_seg = 0, n_prim_mesh_seg = 0, n_hits = 0, n_samples = 0, n_iter = 0;
    unsigned n_bvh_nodes = 0, n_bvh_tests = 0;                          // per lane
    unsigned long long cyc_gen = 0, cyc_trace = 0, cyc_shade = 0, cyc_mesh = 0;
    const unsigned long long t_start = wall_clock64();
    const int count0 = fp.class_count[0], count1 = fp.class_count[1], count2 = fp.class_count[2];
    const int per_tile0 = (fp.n_passes + fp.class_ppi[0] - 1) / fp.class_ppi[0], per_tile1 = (fp.n_passes + fp.class_ppi[1] - 1) / fp.class_ppi[1],
              per_tile2 = (fp.n_passes + fp.class_ppi[2] - 1) / fp.class_ppi[2];
    const int items0 = count0 * per_tile0, items1 = count1 * per_tile1;
    const int n_items = items0 + items1 + count2 * per_tile2;
    for (;;) {
        // every wave leaves through this test once the queue is drained
        unsigned fetched = 0;
        if (lane == 0) fetched = atomicAdd(fp.queue, 1u);
        const int item = (int)__builtin_amdgcn_readfirstlane(fetched);
        if (item >= n_items) break;
        // item -> (class, tile rank, pass chunk): classes are contiguous in rank and in item index
        int rel = item, rank0 = 0, cls = 0;
        if (rel >= items0) { rel -= items0; rank0 = count0; cls = 1; if (rel >= items1) { rel -= items1; rank0 += count1; cls = 2; } }
        const int per_tile = cls == 0 ? per_tile0 : cls == 1 ? per_tile1 : per_tile2;
        const int ppi = fp.class_ppi[cls];
        const int chunk = rel % per_tile;
        const int tile = (int)fp.order[rank0 + rel / per_tile];        // wave-uniform: scalar load
        int x, y;
        const bool valid = tile_pixel(fp, tile, lane, x, y);
        const int k_end = min(fp.n_passes, (chunk + 1) * ppi);
        Rgb *rad = fp.radiance + (size_t)y * fp.width + x;
        int k = chunk * ppi;                                           // pass index relative to fp.first_pass
        bool fresh = true;                                             // lane needs the primary ray of pass k
        PathState ps;
        ps.o = ps.d = ps.ray_color = ps.incoming = ps.emitted = V(0, 0, 0); ps.rng = 0; ps.bounces = 0;
        for (;;) {
            const bool alive = valid && k < k_end;
            if (!wave_any(alive)) break;
            n_iter++;
            PT_TICK(c0);
            if (alive && fresh) {
                ps.rng = primary_ray(fp, sc.sh, x, y, fp.first_pass + k, ps.o, ps.d);
                ps.ray_color = V(1, 1, 1); ps.incoming = V(0, 0, 0); ps.emitted = V(0, 0, 0); ps.bounces = 0;
                fresh = false;
            }
            n_samples += __popcll(wave_ballot(alive && ps.bounces == 0));
            PT_TICK(c1);
            Hit h = calculate_ray<SLOTS>(sc, st, lane, alive, ps.bounces == 0, ps.o, ps.d, n_seg, n_mesh_seg, n_prim_mesh_seg, n_bvh_nodes, n_bvh_tests, cyc_mesh);
            PT_TICK(c2);
            const bool hit = h.id >= 0 && h.t < kInfinity;
            n_hits += __popcll(wave_ballot(alive && hit && !fp.debug_normal));
            if (alive) {
                bool done;
                v3 colour;
                if (hit && fp.debug_normal) {
                    colour = h.normal; done = true;                     // rt.comp:274
                } else {
                    if (hit) {
                        shade_hit(sc, h, ps);
                        done = ps.bounces >= max_bounces;
                    } else {
                        ps.bounces++;
                        ps.emitted = ps.emitted + V(sc.sh.sky_emit[0], sc.sh.sky_emit[1], sc.sh.sky_emit[2]);   // skyColor * skyIntensity (rt.comp:315)
                        ps.ray_color = ps.ray_color * V(sc.sh.sky[0], sc.sh.sky[1], sc.sh.sky[2]);              // rt.comp:316
                        ps.incoming = ps.incoming + ps.emitted * ps.ray_color;
                        done = true;                                    // break (rt.comp:318)
                    }
                    const float nb = (float)ps.bounces;
                    colour = V(ps.incoming.x / nb, ps.incoming.y / nb, ps.incoming.z / nb);   // rt.comp:322
                    if (sc.sh.flags & kShSqrtOut) colour = V(sqrtf(colour.x), sqrtf(colour.y), sqrtf(colour.z));   // rt.frag:320
                }
                if (done) {
                    rad[(size_t)k * plane] = Rgb{colour.x, colour.y, colour.z};
                    k++;
                    fresh = true;
                }
            }
            PT_TICK(c3);
            PT_ACC(cyc_gen, c0, c1); PT_ACC(cyc_trace, c1, c2); PT_ACC(cyc_shade, c2, c3);
        }
    }
    if constexpr (Store::is_bvh) {
        if (fp.counters) {
            for (int off = 32; off > 0; off >>= 1) { n_bvh_nodes += __shfl_xor(n_bvh_nodes, off); n_bvh_tests += __shfl_xor(n_bvh_tests, off); }
            if (lane == 0) { atomicAdd(&fp.counters[10], (unsigned long long)n_bvh_nodes); atomicAdd(&fp.counters[11], (unsigned long long)n_bvh_tests); }
        }
    }
    if (lane == 0 && fp.counters) {
        atomicAdd(&fp.counters[0], (unsigned long long)n_samples);
        atomicAdd(&fp.counters[1], (unsigned long long)n_seg);
        atomicAdd(&fp.counters[2], (unsigned long long)n_mesh_seg);
        if constexpr (!Store::is_bvh) atomicAdd(&fp.counters[3], (unsigned long long)n_mesh_seg * sc.n_chunks * (kWave * SLOTS));
        atomicAdd(&fp.counters[4], (unsigned long long)n_hits);
        atomicAdd(&fp.counters[5], (unsigned long long)n_iter);
        atomicAdd(&fp.counters[6], (unsigned long long)n_prim_mesh_seg);
        if (PT_PHASE_TIMERS) { atomicAdd(&fp.counters[12], cyc_gen); atomicAdd(&fp.counters[13], cyc_trace); atomicAdd(&fp.counters[14], cyc_shade); atomicAdd(&fp.counters[15], cyc_mesh); }
        const unsigned long long span = wall_clock64() - t_start;
        atomicAdd(&fp.counters[7], 1ull);
        atomicAdd(&fp.counters[8], span);
        atomicMax(&fp.counters[9], span);
    }
}


// ---- wave-uniform operands through LDS ----------------------------------------------------------------------------------
// On gfx950 a VALU instruction with an SGPR source issues in 4 SIMD cycles, the same instruction on VGPRs in 2
// (profiles/valu_mix_mi355x.txt).  The scene triangles (SSBO 5), the root AABB and the camera are the same for every lane,
// so the compiler keeps them in SGPRs (s_load) -- and every multiply, subtract and add of the per-lane phases that touches
// them runs at half rate (27 of the 52 instructions of one scene-triangle test).  The traversal kernels therefore stage
// these few values in LDS once per workgroup and read them back with ds_read_b128 at a wave-uniform address (a broadcast:
// no bank conflict, no VALU slot): they arrive in VGPRs.  Layout (float4): [0, 3n) scene triangles (p1 | -, e1, e2),
// then bmin, bmax, (iv0 iv1 iv2 iv3), (iv4 iv5 iv6 iv7), (iv8 cam.x cam.y cam.z).
// In front of them, always: the shading parameters that are read once per path or per pixel (sky, jitter scale, origin
// offset, pixel-centre offset) -- as kernel arguments they would sit in SGPRs through the traversal loops, where the
// kernel already spills; here they cost one ds_read at the point of use.  (sky | jitter_scale), (sky_emit | origin_offset),
// (pix_off, sph_k, sph_rmin, sph_reach), then the box of all spheres (min, max).
constexpr int kShadeLdsVecs = 5;
constexpr int kSceneLdsTris = 16;                                  // scenes with more triangles keep the scalar path
constexpr int kSceneLdsVecs = kShadeLdsVecs + 3 * kSceneLdsTris + 5;
__device__ __forceinline__ bool stage_scene_lds(float4 *dst, const SceneView &sc, const FrameParams &fp, int tid, int n_threads)
{
    if (tid == 0) {
        dst[0] = make_float4(sc.sh.sky[0], sc.sh.sky[1], sc.sh.sky[2], sc.sh.jitter_scale);
        dst[1] = make_float4(sc.sh.sky_emit[0], sc.sh.sky_emit[1], sc.sh.sky_emit[2], sc.sh.origin_offset);
        dst[2] = make_float4(sc.sh.pix_off, sc.sph_k, sc.sph_rmin, sc.sph_reach);
        dst[3] = make_float4(sc.sph_bmin[0], sc.sph_bmin[1], sc.sph_bmin[2], 0.0f);      // the box of all spheres (sphere tree's root)
        dst[4] = make_float4(sc.sph_bmax[0], sc.sph_bmax[1], sc.sph_bmax[2], 0.0f);
    }
    dst += kShadeLdsVecs;
    if (sc.n_tris > kSceneLdsTris) return false;
    for (int i = tid; i < 3 * sc.n_tris; i += n_threads) dst[i] = sc.prim_tab[4 * (sc.n_mesh + i / 3) + i % 3];
    if (tid == 0) {
        float4 *c = dst + 3 * sc.n_tris;
        c[0] = make_float4(sc.bmin[0], sc.bmin[1], sc.bmin[2], 0.0f);
        c[1] = make_float4(sc.bmax[0], sc.bmax[1], sc.bmax[2], 0.0f);
        c[2] = make_float4(fp.iv[0], fp.iv[1], fp.iv[2], fp.iv[3]);
        c[3] = make_float4(fp.iv[4], fp.iv[5], fp.iv[6], fp.iv[7]);
        c[4] = make_float4(fp.iv[8], fp.cam[0], fp.cam[1], fp.cam[2]);
    }
    return true;
}
__device__ __forceinline__ float4 lds_f4(lds_cf4v *p, int i) { const f4v v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }

// scene_closest_tab with the triangle records read from the LDS copy (VGPR operands); spheres as before
__device__ __forceinline__ void scene_closest_lds(const SceneView &sc, lds_cf4v *scene, v3 o, v3 d, float &best_t, int &best_id, bool with_spheres, bool xyz)
{
    for (int i = 0; i < sc.n_tris; i++) {
        const float4 p1 = lds_f4(scene, 3 * i), e1 = lds_f4(scene, 3 * i + 1), e2 = lds_f4(scene, 3 * i + 2);
        float t, det;
        bool has = hit_triangle(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, sc.sh.hit_limit);
        if (has && t < best_t) { best_t = t; best_id = sc.n_mesh + i; }
    }
    if (!with_spheres) return;
    for (int i = 0; i < sc.n_sph; i++) {
        const float4 s = uniform_load(sc.sph, i);
        float t; bool front;
        bool has = hit_sphere(V(s.x, s.y, s.z), s.w, o, d, t, front, sc.sh.hit_limit, xyz);
        if (has && t < best_t) { best_t = t; best_id = sc.n_mesh + sc.n_tris + i; }
    }
}

// ------------------------------------------------------------------------------------------------
// BVH body: the same persistent waves, work items and persistent lanes as trace_items_body, but a lane's path is a
// small state machine so that rays which need the mesh can WAIT for each other:
//   FRESH -> (primary ray) -> TRACE -> (root-AABB gate + scene primitives) -> READY, or MESH if the gate passed
//   MESH  -> (BVH traversal, resumable)                                       -> READY
//   READY -> (hit record, shading / sky, store)                               -> TRACE (next bounce) | FRESH (next pass) | DEAD
// Every loop iteration runs the resolve / generate / trace phases for the lanes that are in those states.  The
// traversal phase runs only when enough lanes are parked in MESH (fp.park_high), when they have waited fp.park_wait
// iterations, or when nothing else is left to do; inside it every step is either one node visit or one triangle test for
// the lanes that want that kind (majority vote), and it returns early once three quarters of its rays are finished,
// leaving the rest parked WITH their traversal state (node, stack, best hit) to be resumed beside the next batch.  So the
// traversal runs with most lanes busy instead of at the occupancy of whichever iteration a ray happened to be born in.
// Scheduling only: a sample is a pure function of (pixel, pass); mesh-before-scene tie order is kept by the merge rule
// in `finish`: a scene primitive wins only with a strictly smaller t (rt.comp:213-230 scans the mesh first).
// (t, id) < (best_t, best_id) in lexicographic order -- what the reference's in-order strict `<` scan keeps -- without
// short-circuit evaluation, so that the update around it compiles to selects, not to a nest of exec-mask branches (16
// instructions and three branches per triangle before; config 2 kernel 6.23 -> 6.14 ms).  Not as one integer comparison
// of the bit patterns: a sphere counts as hit down to t1 = -HIT_LIMIT (rt.comp:142), so t can be negative or -0 (tried:
// 46 pixels of config 5 differed).
__device__ __forceinline__ bool closer_hit(float t, int id, float best_t, int best_id)
{
    return (t < best_t) | ((t == best_t) & (id < best_id));
}

enum : int { S_DEAD = 0, S_FRESH = 1, S_TRACE = 2, S_MESH = 3, S_READY = 4, S_DIR = 5 };

// SPH: the scene has a sphere tree (compile-time, so scenes without spheres carry none of its code or registers).
// SHORT: launches of a few passes (interactive use): items are dealt round-robin to the waves instead of pulled from the
// queue, every item covers all passes, and the lane keeps the running mean in registers (no workspace, no pt_accumulate).
// GEN: everything in Shading is read from the arguments.  Otherwise (the kernels pin these fields of their by-value copy
// before calling, so the compiler folds them) the kSh* flags -- the places where rt.frag differs from rt.comp in the FORM
// of an expression -- and the hit limit with its box slack are rt.comp's at compile time: the flags' selects and branches
// vanish, and the two limits, which are compared in every step of the traversal loop, do not take two more SGPRs there
// (measured: +0.10 ms of 6.2 on config 2, the loop already spills).  The other numeric parameters are arguments either
// way.  launch_trace picks GEN whenever flags, hit limit or slack differ from rt.comp's.
template <bool SPH, bool SHORT, bool GEN, class Store>
__device__ __forceinline__ void trace_items_body_bvh(const SceneView &sc, const FrameParams &fp, Store &st, int lane)
{
    const int sh_flags = sc.sh.flags;
    const bool sh_xyz = (sh_flags & kShSumXyz) != 0;
    const int sph_root = SPH ? sc.sph_root : -1;
    const int max_bounces = fp.debug_normal ? 1 : fp.max_bounces;      // rt.comp:23
    const unsigned plane = (unsigned)(fp.width * fp.height);          // sample index k * plane + pixel < 2^32 (launch_trace bounds the batch)
    unsigned n_seg = 0, n_mesh_seg = 0, n_prim_mesh_seg = 0, n_hits = 0, n_samples = 0, n_iter = 0;
    unsigned n_bvh_nodes = 0, n_bvh_tests = 0, n_sph_tests = 0;         // per lane
    unsigned long long cyc_gen = 0, cyc_trace = 0, cyc_shade = 0, cyc_mesh = 0;
    const unsigned long long t_start = wall_clock64();
    unsigned *stk = st.stack;
    // Lanes are scheduled one by one: a lane that has finished its pixel takes the next unassigned pixel of the wave's
    // current work item (a tile of 64 px x a range of passes), and the wave fetches another item from the queue when that
    // one is used up.  So a wave keeps ~64 live paths whatever their lengths, and lanes parked for the mesh can wait for
    // company without idling the others.
    int t_valid = 0, t_next = 0, t_tile = 0, t_k0 = 0, t_kend = 0;          // wave-uniform cursor into the current item
    const int static_stride = (int)(gridDim.x * (blockDim.x / kWave));
    int static_next = (int)(blockIdx.x + gridDim.x * (threadIdx.x / kWave));   // waves of one workgroup take items far apart in the list
    bool more = true;                                                       // the queue may still hold items
    int x = 0, y = 0, k = 0, k_end = 0;                                     // this lane's pixel and pass range
    // fused accumulation (SHORT: the lane traces ALL passes of its pixel in order): the running mean lives here
    v3 acc = V(0, 0, 0);
    float acc_w = 0.0f;
    uint32_t pix_seed = 0;                                                  // pixelIndex (rt.comp:16); the rng seed is pass * pixelIndex
    float pix_x = 0.0f, pix_y = 0.0f;                                       // pixelPos (rt.comp:14,17)
    // a continuing path between the resolve and direction phases: hit point, normal, refraction index, mix weights
    v3 hp = V(0, 0, 0), nrm = V(0, 0, 0);
    float eta = 1.0f, mix_a = 0.0f, mix_b = 0.0f;
    bool refr = false;
    int state = S_DEAD;
    PathState ps;
    ps.o = ps.d = ps.ray_color = ps.incoming = ps.emitted = V(0, 0, 0); ps.rng = 0; ps.bounces = 0;
    // closest hit: scene primitives (s_t, s_id); mesh traversal / final result (best_t, best_id)
    float s_t = kInfinity, best_t = kInfinity;
    int s_id = -1, best_id = -1;
    int best_rec = -1;                // leaf-order index of mesh triangle best_id (Store::kLeafRecords), -1: not from a leaf
    // resumable traversal state (see the traversal phase): position `cur`, stack height `sp` (slot 0 = kTravDone for good)
    int cur = kTravDone, sp = 1;
    stk[0] = (unsigned)kTravDone;
    v3 inv = V(0, 0, 0);
    float delta = 0.0f;
    int wait = 0;
    {
        for (;;) {
            // ---- refill dead lanes with the next pixels
            unsigned long long dead = wave_ballot(state == S_DEAD);
            if (more && dead && (__popcll(dead) >= fp.refill_min || !(wave_ballot(state == S_FRESH) | wave_ballot(state == S_TRACE) | wave_ballot(state == S_READY)))) {
                while (dead) {
                    if (t_next >= t_valid) {
                        int item;
                        if (SHORT) {
                            // wave w takes items w, w + waves, ... of the heaviest-first list (no queue: one device-wide
                            // atomic costs ~11 ns serialised, and a 1-spp frame has only ~8 items per wave)
                            item = static_next; static_next += static_stride;
                        } else {
                            // every wave stops fetching once the queue is drained
                            unsigned fetched = 0;
                            if (lane == 0) fetched = atomicAdd(fp.queue, 1u);
                            item = (int)__builtin_amdgcn_readfirstlane(fetched);
                        }
                        // the class sizes are read here, per item, not kept in registers across the loop: a fetch is rare
                        // (and costs a device-wide atomic anyway), while eight more live SGPRs spill in every iteration
                        const volatile int *cc = fp.class_count;
                        const int count0 = __builtin_amdgcn_readfirstlane(cc[0]), count1 = __builtin_amdgcn_readfirstlane(cc[1]),
                                  count2 = __builtin_amdgcn_readfirstlane(cc[2]);
                        const int items0 = count0 * fp.per_tile[0], items1 = count1 * fp.per_tile[1];
                        const int n_items = items0 + items1 + count2 * fp.per_tile[2];
                        if (item >= n_items) { more = false; break; }
                        // item -> (class, tile rank, pass chunk): classes are contiguous in rank and in item index
                        int rel = item, rank0 = 0, cls = 0;
                        if (rel >= items0) { rel -= items0; rank0 = count0; cls = 1; if (rel >= items1) { rel -= items1; rank0 += count1; cls = 2; } }
                        const int per_tile = fp.per_tile[cls];
                        const int ppi = fp.class_ppi[cls];
                        const int chunk = rel % per_tile;
                        const int tile = (int)fp.order[rank0 + rel / per_tile];        // wave-uniform: scalar load
                        t_tile = tile;
                        t_k0 = chunk * ppi; t_kend = min(fp.n_passes, (chunk + 1) * ppi);
                        t_valid = tile_pixels(fp, tile); t_next = 0;
                    }
                    const int n_dead = __popcll(dead), room = t_valid - t_next;
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(dead >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dead, 0));   // dead lanes below this one
                    if (state == S_DEAD && rank < room) {
                        tile_pixel(fp, t_tile, t_next + rank, x, y);
                        k = t_k0; k_end = t_kend;
                        const float W = (float)fp.width, H = (float)fp.height;
                        const float pix_off = lds_f4(st.shade, 2).x;
                        const float fx = (float)x + pix_off, fy = (float)y + pix_off;   // rt.comp:12 | rt.frag:14 (x + 0.5 is exact)
                        pix_seed = pixel_index(fx, fy, W, H);
                        pix_x = (((fx / W) - 0.5f) * W) / H;            // rt.comp:14,17
                        pix_y = (fy / H) - 0.5f;
                        if (SHORT) {
                            const float4 prev = reinterpret_cast<const float4 *>(fp.accum)[(size_t)y * fp.width + x];
                            acc = V(prev.x, prev.y, prev.z); acc_w = prev.w;
                        }
                        state = S_FRESH;
                    }
                    t_next += min(n_dead, room);
                    dead = wave_ballot(state == S_DEAD);
                }
            }
            const unsigned long long m_mesh = wave_ballot(state == S_MESH);
            // one vote per comparison, combined as masks: a vote on a compound condition costs a v_cndmask + v_cmp pair extra
            const unsigned long long m_other = wave_ballot(state == S_FRESH) | wave_ballot(state == S_TRACE) | wave_ballot(state == S_READY);   // S_DIR never survives an iteration
            if (!(m_mesh | m_other)) break;      // nothing alive and (refill ran) nothing left in the queue
            n_iter++;
            const int n_mesh = __popcll(m_mesh), n_other = __popcll(m_other);
            PT_TICK(c0);
            if (n_mesh && (n_mesh >= fp.park_high || n_other == 0 || wait >= fp.park_wait)) {
                // ---- traversal phase.  A lane's whole position is ONE register: cur >= 0 inner node to visit; cur < 0 the code
                // of the leaf it is in (~cur = flag | first << 3 | count - 1, rewritten as its primitives are used up); cur ==
                // kTravDone not traversing.  Slot 0 of every lane's stack holds kTravDone for good, so "pop" needs no empty test,
                // and both kinds of step read the top of the stack speculatively together with their node / primitive data.
                // The step bodies are written without branches (selects, one unconditional store to the free slot above the
                // stack): on this kernel a scalar instruction costs as much issue time as a vector one (profiles/insitu_cost_*),
                // and the nest of exec-mask branches this replaces was 45 of them per node visit.
                wait = 0;
                PT_COUNT(cyc_gen, 1); PT_COUNT(cyc_mesh, n_mesh);
                const int exit_at = n_other > 0 ? (n_mesh * fp.park_exit) >> 3 : 0;
                const v3 nlo = V(-((ps.o.x + delta) * inv.x), -((ps.o.y + delta) * inv.y), -((ps.o.z + delta) * inv.z));
                const v3 nhi = V(-((ps.o.x - delta) * inv.x), -((ps.o.y - delta) * inv.y), -((ps.o.z - delta) * inv.z));
#if PT_SLAB_BY_SIGN
                // the wide node step reads a node's planes sorted by the ray's direction: the addend that goes with the plane the
                // ray enters through (min plane, grown outwards: nlo -- or max plane: nhi) and with the one it leaves through
                const v3 nn = V(inv.x < 0.0f ? nhi.x : nlo.x, inv.y < 0.0f ? nhi.y : nlo.y, inv.z < 0.0f ? nhi.z : nlo.z);
                const v3 nf = V(inv.x < 0.0f ? nlo.x : nhi.x, inv.y < 0.0f ? nlo.y : nhi.y, inv.z < 0.0f ? nlo.z : nhi.z);
#endif
                // One leaf's worth of primitives, read from `code_neg` (a leaf code): a sphere, or up to two triangles (independent
                // tests: the second hides the first's latencies).  Returns how many were used.
                auto test_leaf = [&](int code_neg) -> int {
                    const int code = ~code_neg, first = (code & (kSphereLeaf - 1)) >> 3, cnt = (code & 7) + 1;
                    if (SPH && (code & kSphereLeaf) != 0) {
                        // one sphere of a sphere-tree leaf per step (rt.comp:126-146), canonical id after the mesh and scene
                        // triangles.  Two per step, like the triangles, was measured and is slower (config 5: 36.5 vs 35.1 ms): the
                        // test is dear (a square root and two divisions) and in a leaf of one or three the second is wasted
                        float4 cr;
                        int si;
                        st.sphere(first, cr, si);
                        float t; bool front;
                        const bool has = hit_sphere(V(cr.x, cr.y, cr.z), cr.w, ps.o, ps.d, t, front, sc.sh.hit_limit, sh_xyz);
                        const int id = sc.n_mesh + sc.n_tris + si;
                        n_sph_tests++;
                        const bool take = has & closer_hit(t, id, best_t, best_id);
                        best_t = take ? t : best_t; best_id = take ? id : best_id;
                        return 1;
                    }
                    const bool two = cnt >= 2;
                    float4 p1, e1, e2, q1, f1, f2;
                    st.tri(first, p1, e1, e2);
                    st.tri(two ? first + 1 : first, q1, f1, f2);
                    float t, det, t2, det2;
                    bool bad = false;
                    bool has = hit_triangle_t<true>(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), ps.o, ps.d, t, det, bad, sc.sh.hit_limit);
                    bool has2 = hit_triangle_t<true>(V(q1.x, q1.y, q1.z), V(f1.x, f1.y, f1.z), V(f2.x, f2.y, f2.z), ps.o, ps.d, t2, det2, bad, sc.sh.hit_limit);
                    if (__builtin_expect(bad, 0)) {
                        has = hit_triangle_ieee(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), ps.o, ps.d, t, det, sc.sh.hit_limit);
                        has2 = hit_triangle_ieee(V(q1.x, q1.y, q1.z), V(f1.x, f1.y, f1.z), V(f2.x, f2.y, f2.z), ps.o, ps.d, t2, det2, sc.sh.hit_limit);
                    }
                    const int id = __float_as_int(p1.w), id2 = __float_as_int(q1.w);
                    n_bvh_tests += two ? 2 : 1;
                    const bool take = has & closer_hit(t, id, best_t, best_id);
                    best_t = take ? t : best_t; best_id = take ? id : best_id;
                    const bool take2 = (two & has2) & closer_hit(t2, id2, best_t, best_id);
                    best_t = take2 ? t2 : best_t; best_id = take2 ? id2 : best_id;
                    if constexpr (Store::kLeafRecords) { best_rec = take ? first : best_rec; best_rec = take2 ? first + 1 : best_rec; }
                    return two ? 2 : 1;
                };
                for (;;) {
                    const unsigned long long in_mesh = wave_ballot(cur != kTravDone);
                    const int n = __popcll(in_mesh);
                    if (n <= exit_at) break;
                    const int n_node = __popcll(wave_ballot(cur >= 0));
                    int new_cur = cur;                     // lanes of the other kind keep their position
                    if (2 * n_node >= n) {
                        PT_COUNT(cyc_trace, 1);
                        if (kBvhWide && cur >= 0) {
                            // four-wide node: the nearest child that is hit is visited next, the other hit children go on the
                            // stack -- four unconditional stores, each kept (the height moves on) only if its child counts
                            int c0, c1, c2, c3;
                            const float lim = fmaf(best_t, 1.001f, delta);
                            float e0, e1, e2, e3;
                            bool h0, h1, h2, h3;
#if PT_SLAB_BY_SIGN
                            // planes sorted by the ray's direction: entry = the largest of the three near-plane distances, exit = the
                            // smallest of the three far-plane ones; no min / max per axis (they issue at the slow VALU rate)
                            float4 nx, ny, nz, fx, fy, fz;
                            v3 pm = inv, pan = nn, paf = nf;        // t = plane * pm + pan | paf (the quantised store folds its decoding in)
                            st.node4_dir(cur, __float_as_uint(inv.x) >> 31, __float_as_uint(inv.y) >> 31, __float_as_uint(inv.z) >> 31, nx, ny, nz, fx, fy, fz, c0, c1, c2, c3, pm, pan, paf);
                            const int top = (int)stk[(sp - 1) * kWave];
                            n_bvh_nodes++;
#define PT_SLOT(k, h, e, extra) { const float a0 = __builtin_fmaxf(__builtin_fmaxf(fmaf(nx.k, pm.x, pan.x), fmaf(ny.k, pm.y, pan.y)), fmaf(nz.k, pm.z, pan.z)); \
                                  const float a1 = __builtin_fminf(__builtin_fminf(fmaf(fx.k, pm.x, paf.x), fmaf(fy.k, pm.y, paf.y)), fmaf(fz.k, pm.z, paf.z)) * 1.000001f; \
                                  h = a0 <= a1 && a1 >= sc.sh.box_slack && a0 <= lim && (extra); e = a0; }
                            PT_SLOT(x, h0, e0, true)      // slots 0 and 1 are always in use
                            PT_SLOT(y, h1, e1, true)
                            PT_SLOT(z, h2, e2, (!Store::kChildTest || c2 != kTravDone))
                            PT_SLOT(w, h3, e3, (!Store::kChildTest || c3 != kTravDone))
#undef PT_SLOT
#else
                            float4 mnx, mny, mnz, mxx, mxy, mxz;
                            st.node4(cur, mnx, mny, mnz, mxx, mxy, mxz, c0, c1, c2, c3);
                            const int top = (int)stk[(sp - 1) * kWave];
                            n_bvh_nodes++;
                            float a0, a1;
                            slab_test_fma(mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, nlo, nhi, inv, a0, a1);
                            h0 = a0 <= a1 && a1 >= sc.sh.box_slack && a0 <= lim;      // slots 0 and 1 are always in use
                            e0 = a0;
                            slab_test_fma(mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, nlo, nhi, inv, a0, a1);
                            h1 = a0 <= a1 && a1 >= sc.sh.box_slack && a0 <= lim;
                            e1 = a0;
                            slab_test_fma(mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, nlo, nhi, inv, a0, a1);
                            h2 = a0 <= a1 && a1 >= sc.sh.box_slack && a0 <= lim && (!Store::kChildTest || c2 != kTravDone);
                            e2 = a0;
                            slab_test_fma(mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, nlo, nhi, inv, a0, a1);
                            h3 = a0 <= a1 && a1 >= sc.sh.box_slack && a0 <= lim && (!Store::kChildTest || c3 != kTravDone);
                            e3 = a0;
#endif
                            // nearest hit child: a running minimum over the slots; b_k = slot k took the lead from the slots before it
                            float et = h0 ? e0 : kInfinity * 4.0f;
                            int ec = c0;
                            const bool b1 = h1 && e1 < et; et = b1 ? e1 : et; ec = b1 ? c1 : ec;
                            const bool b2 = h2 && e2 < et; et = b2 ? e2 : et; ec = b2 ? c2 : ec;
                            const bool b3 = h3 && e3 < et; ec = b3 ? c3 : ec;
                            const bool any = h0 | h1 | h2 | h3;
                            // a hit child is pushed unless it ended up the nearest (slot k is the nearest iff b_k and no later b)
                            const bool p0 = h0 & (b1 | b2 | b3), p1 = h1 & (!b1 | b2 | b3), p2 = h2 & (!b2 | b3), p3 = h3 & !b3;
                            int s = sp;
                            stk[s * kWave] = (unsigned)c0; s += p0 ? 1 : 0;
                            stk[s * kWave] = (unsigned)c1; s += p1 ? 1 : 0;
                            stk[s * kWave] = (unsigned)c2; s += p2 ? 1 : 0;
                            stk[s * kWave] = (unsigned)c3; s += p3 ? 1 : 0;
                            new_cur = any ? ec : top;
                            sp = any ? s : sp - 1;
                        } else if (cur >= 0) {
                            float4 a, b, c;
                            int kl, kr;
                            st.node(cur, a, b, c, kl, kr);
                            const int top = (int)stk[(sp - 1) * kWave];
                            n_bvh_nodes++;
                            float l0, l1, r0, r1;
                            slab_test_fma(a.x, a.y, a.z, a.w, b.x, b.y, nlo, nhi, inv, l0, l1);
                            slab_test_fma(b.z, b.w, c.x, c.y, c.z, c.w, nlo, nhi, inv, r0, r1);
                            const float lim = fmaf(best_t, 1.001f, delta);
                            // a sphere counts as hit down to t = -HIT_LIMIT (rt.comp:146), hence the slack on the far side
                            const bool hl = l0 <= l1 && l1 >= sc.sh.box_slack && l0 <= lim, hr = r0 <= r1 && r1 >= sc.sh.box_slack && r0 <= lim;
                            const bool take_left = hl && (!hr || l0 <= r0);
                            const int near = take_left ? kl : kr, far = take_left ? kr : kl;
                            stk[sp * kWave] = (unsigned)far;            // stays on the stack only if both children are hit
                            const bool both = hl && hr, any = hl || hr;
                            new_cur = any ? near : top;
                            sp += both ? 1 : (any ? 0 : -1);
                        }
                    } else {
                        PT_COUNT(cyc_shade, 1);
                        if (cur < 0 && cur != kTravDone) {
                            const int cnt = ((~cur) & 7) + 1;
                            const int top = (int)stk[(sp - 1) * kWave];
                            const int used = test_leaf(cur);
                            // primitives left: the leaf code moves on by `used` (first += used, count -= used: ~code falls by
                            // 8 * used - used); none left: pop
                            const bool more_left = !kOneStepLeaves && cnt > used;
                            new_cur = more_left ? cur - 7 * used : top;
                            sp -= more_left ? 0 : 1;
                        }
                    }
                    const bool fin = new_cur == kTravDone && cur != kTravDone;      // popped the bottom of the stack
                    cur = new_cur;
                    state = fin ? S_READY : state;
                }
            } else if (n_mesh) {
                wait++;
            }
            PT_TICK(c1);
            PT_ACC(cyc_mesh, c0, c1);
            // ---- resolve phase (rt.comp:273-322), up to the point where a continuing path needs its random direction
            if (wave_any(state == S_READY)) {
                if (!fp.debug_normal) n_hits += __popcll(wave_ballot(state == S_READY) & wave_ballot(best_id >= 0) & wave_ballot(best_t < kInfinity));
                if (state == S_READY) {
                    Hit h;
                    h.id = best_id; h.t = best_t; h.front = false; h.normal = V(0, 0, 0); h.mat = 0;
                    finish_hit_tab(sc, ps.o, ps.d, h, sh_xyz, (Store::kLeafRecords && best_id >= 0 && best_id < sc.n_mesh) ? best_rec : -1);
                    const bool hit = h.id >= 0 && h.t < kInfinity;
                    bool done = true;
                    v3 colour;
                    if (hit && fp.debug_normal) {
                        colour = h.normal;                              // rt.comp:274
                    } else {
                        if (hit) {
                            // rt.comp:276-311; the draws keep the shader's order: specular, refraction, then the direction
                            ps.bounces++;
                            const PtMaterial *m = &sc.mats[h.mat];
                            const float m_ior = m->ior, m_smooth = m->smoothness;
                            // FresnelReflectAmount(n1 = ior, n2 = m_ior) (rt.comp:235-254, :283): its two divisions and the 1/ior of
                            // rt.comp:281 depend on the material only -- taken from the host's table (same IEEE operations)
                            const float4 mf = sc.mat_fres[h.mat];
                            const float ior = h.front ? mf.x : m_ior;
                            const bool is_spec = m->specularChance > random_value(ps.rng);
                            // n1 = ior (rt.comp:283) or 1.0 (rt.frag:283): the table follows the mode (pt_api.hip fresnel_table)
                            const float fres = fresnel_reflect_amount_pre(h.front ? mf.y : mf.w, ((sh_flags & kShN1One) ? 1.0f : ior) > m_ior, h.front ? mf.z : mf.x,
                                                                          h.normal, ps.d, 1.0f - m->refractionAmount);
                            const bool is_refr = fres < random_value(ps.rng);
                            const v3 base = V(m->baseColor[0], m->baseColor[1], m->baseColor[2]);
                            const v3 spec = V(m->specularColor[0], m->specularColor[1], m->specularColor[2]);
                            ps.emitted = ps.emitted + V(m->emissionColor[0], m->emissionColor[1], m->emissionColor[2]) * m->emissionStrength;
                            ps.ray_color = ps.ray_color * select3(is_refr, base, select3(is_spec, spec, base));
                            ps.incoming = ps.incoming + ps.emitted * ps.ray_color;
                            done = ps.bounces >= max_bounces;
                            if (!done) {
                                // the new direction (rt.comp:296-305) is finished in the direction phase; at the last
                                // bounce the shader computes one too, but nothing reads it
                                hp = ps.o + ps.d * h.t;                 // hitPoint (rt.comp:175 / :144)
                                nrm = h.normal; eta = ior; refr = is_refr;
                                mix_a = is_spec ? m->specularSmoothness : m_smooth; mix_b = m_smooth;
                                state = S_DIR;
                            }
                        } else {
                            ps.bounces++;
                            const float4 sky = lds_f4(st.shade, 0), sky_emit = lds_f4(st.shade, 1);
                            ps.emitted = ps.emitted + V(sky_emit.x, sky_emit.y, sky_emit.z);    // skyColor * skyIntensity (rt.comp:315)
                            ps.ray_color = ps.ray_color * V(sky.x, sky.y, sky.z);               // rt.comp:316
                            ps.incoming = ps.incoming + ps.emitted * ps.ray_color;
                        }
                        const float nb = (float)ps.bounces;
                        colour = div_small_int(ps.incoming, nb);   // rt.comp:322
                        if (sh_flags & kShSqrtOut) colour = V(exact_sqrt(colour.x), exact_sqrt(colour.y), exact_sqrt(colour.z));   // rt.frag:320
                    }
                    if (done) {
                        if (SHORT) {
                            // rt.comp:337-339 in pass order, in registers: accum = mix(accum, colour, 1/pass)
                            if (fp.sum_mode) acc = acc + colour;
                            else acc = mix3(acc, colour, exact_rcp((float)(fp.first_pass + k)));
                        } else {
                            fp.radiance[(unsigned)k * plane + (unsigned)(y * fp.width + x)] = Rgb{colour.x, colour.y, colour.z};
                        }
                        k++;
                        state = k < k_end ? S_FRESH : S_DEAD;
                        if (SHORT && k == k_end)
                            reinterpret_cast<float4 *>(fp.accum)[(size_t)y * fp.width + x] =
                                make_float4(acc.x, acc.y, acc.z, fp.sum_mode ? acc_w + (float)fp.n_passes : 1.0f);
                    }
                }
            }
            PT_TICK(c2);
            PT_ACC(cyc_shade, c1, c2);
            // ---- direction phase: RandomDirection (rt.comp:85-90) once for both of its callers -- the jitter of a new
            // primary ray (rt.comp:326-333) and the bounce direction of a continuing path (rt.comp:296-305)
            if (wave_ballot(state == S_FRESH) | wave_ballot(state == S_DIR)) {
#ifdef PT_DUMMY_VALU
#include "../../tools/insitu_dummy.inc"      // diagnostic builds only (tools/insitu_cost.sh): extra instructions of one form, priced in situ
#endif
                n_samples += __popcll(wave_ballot(state == S_FRESH));
                if (state == S_FRESH || state == S_DIR) {
                    const bool fresh = state == S_FRESH;
                    if (fresh) ps.rng = (uint32_t)(fp.first_pass + k) * pix_seed;       // rt.comp:327
                    const float gx = random_value_nd(ps.rng), gy = random_value_nd(ps.rng), gz = random_value_nd(ps.rng);
                    const v3 g = V(gx, gy, gz);
                    const float gr = exact_rcp(exact_sqrt(dot(g, g)));        // normalize(v) = v * (1/sqrt(dot(v,v)))
                    if (fresh) {
                        // `RandomDirection(s) * 0.00025` (rt.comp:332) as the reference's compiler folds it
                        const v3 jit = (g * lds_f4(st.shade, 0).w) * gr;
                        const float X = -pix_x + jit.x, Y = pix_y + jit.y;
                        if (st.scene) {     // inverseView and the camera position from the LDS copy: VGPR operands
                            const float4 m0 = lds_f4(st.scene, 3 * sc.n_tris + 2), m1 = lds_f4(st.scene, 3 * sc.n_tris + 3), m2 = lds_f4(st.scene, 3 * sc.n_tris + 4);
                            ps.d = normalize(V((m0.x * X + m1.z) + m0.w * Y, (m0.y * X + m1.w) + m1.x * Y, (m0.z * X + m2.x) + m1.y * Y));
                            ps.o = V(m2.y, m2.z, m2.w);
                        } else {
                            const float *m = fp.iv;
                            ps.d = normalize(V((m[0] * X + m[6]) + m[3] * Y, (m[1] * X + m[7]) + m[4] * Y, (m[2] * X + m[8]) + m[5] * Y));
                            ps.o = V(fp.cam[0], fp.cam[1], fp.cam[2]);
                        }
                        ps.ray_color = V(1, 1, 1); ps.incoming = V(0, 0, 0); ps.emitted = V(0, 0, 0); ps.bounces = 0;
                    } else {
                        const v3 r = g * gr;
                        const v3 hemi = dot(r, nrm) > 0.0f ? r : -r;    // rt.comp:92-95
                        v3 hemi2 = hemi;
                        if (sh_flags & kShTwoDraws) {                        // rt.frag:295: the refraction branch draws its own direction
                            const float hx = random_value_nd(ps.rng), hy = random_value_nd(ps.rng), hz = random_value_nd(ps.rng);
                            const v3 h2 = V(hx, hy, hz);
                            const v3 r2 = h2 * exact_rcp(exact_sqrt(dot(h2, h2)));
                            hemi2 = dot(r2, nrm) > 0.0f ? r2 : -r2;
                        }
                        const v3 a = mix3(hemi, reflect3(ps.d, nrm), mix_a);
                        const v3 b = mix3(-hemi2, refract3(ps.d, nrm, eta), mix_b);
                        ps.d = normalize(select3(refr, b, a));
                        ps.o = hp + ps.d * lds_f4(st.shade, 1).w;       // rt.comp:305
                    }
                    state = S_TRACE;
                }
            }
            PT_TICK(c3);
            PT_ACC(cyc_gen, c2, c3);
            // ---- trace phase: root-AABB gate (rt.comp:213) and scene primitives (rt.comp:219-230); the mesh is deferred
            if (wave_any(state == S_TRACE)) {
                bool gate = false;
                if (state == S_TRACE) {
                    s_t = kInfinity; s_id = -1;
                    if (st.scene) {     // root AABB and scene triangles from the LDS copy: VGPR operands
                        if (sc.n_mesh > 0 && (sh_flags & kShTraceMesh)) {
                            const float4 bn = lds_f4(st.scene, 3 * sc.n_tris), bx = lds_f4(st.scene, 3 * sc.n_tris + 1);
                            gate = hit_aabb(V(bn.x, bn.y, bn.z), V(bx.x, bx.y, bx.z), ps.o, ps.d);
                        }
                        scene_closest_lds(sc, st.scene, ps.o, ps.d, s_t, s_id, sph_root < 0, sh_xyz);
                    } else {
                        if (sc.n_mesh > 0 && (sh_flags & kShTraceMesh)) gate = hit_aabb(V(sc.bmin[0], sc.bmin[1], sc.bmin[2]), V(sc.bmax[0], sc.bmax[1], sc.bmax[2]), ps.o, ps.d);
                        scene_closest_tab(sc, ps.o, ps.d, s_t, s_id, sph_root < 0, sh_xyz);
                    }
                }
                n_seg += __popcll(wave_ballot(state == S_TRACE));
                n_mesh_seg += __popcll(wave_ballot(gate));
                n_prim_mesh_seg += __popcll(wave_ballot(gate) & wave_ballot(ps.bounces == 0));
                if (state == S_TRACE) {
                    best_rec = -1;
                    // the mesh tree (root 0) only behind the gate, like the reference; the sphere tree whenever the ray can
                    // meet the box of all spheres -- the traversal's own conservative slab test, one node visit's worth, here
                    // where it saves the ray a whole turn of parking
                    bool sph_in = false;
                    if (SPH && sph_root >= 0) {
                        inv = V(slab_inv(ps.d.x), slab_inv(ps.d.y), slab_inv(ps.d.z));
                        const float o1 = (fabsf(ps.o.x) + fabsf(ps.o.y)) + fabsf(ps.o.z);
                        // + the reach of the sphere test's rounding (SceneView::sph_k): k D^2 / (sqrt(r^2 + k D^2) + r), D = |o|_1 + reach
                        const float4 sm = lds_f4(st.shade, 2);
                        const float kd2 = sm.y * ((o1 + sm.w) * (o1 + sm.w));
                        delta = fmaf(4e-6f, o1 + sc.bvh_scale, 1.01f * kd2 * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(sm.z, sm.z, kd2)) + sm.z));
                        const v3 nlo = V(-((ps.o.x + delta) * inv.x), -((ps.o.y + delta) * inv.y), -((ps.o.z + delta) * inv.z));
                        const v3 nhi = V(-((ps.o.x - delta) * inv.x), -((ps.o.y - delta) * inv.y), -((ps.o.z - delta) * inv.z));
                        const float4 bn = lds_f4(st.shade, 3), bx = lds_f4(st.shade, 4);
                        float b0, b1;
                        slab_test_fma(bn.x, bn.y, bn.z, bx.x, bx.y, bx.z, nlo, nhi, inv, b0, b1);
                        sph_in = b0 <= b1 && b1 >= sc.sh.box_slack;
                    }
                    if (gate || sph_in) {
                        state = S_MESH;
                        // the traversal starts from the hit among the primitives scanned one by one: the reference's in-order
                        // strict `<` scan returns the lexicographic minimum of (t, canonical id) over all of them, so the order
                        // of the candidates is free -- and subtrees beyond that hit are pruned from the first step
                        sp = 1; best_t = s_t; best_id = s_id;
                        cur = gate ? 0 : sph_root;
                        if (gate && sph_in) { stk[kWave] = (unsigned)sph_root; sp = 2; }
                        // the slab tests only steer the traversal: v_rcp_f32 (1 ulp) is far inside their margins
                        if (!(SPH && sph_root >= 0)) {
                            inv = V(slab_inv(ps.d.x), slab_inv(ps.d.y), slab_inv(ps.d.z));
                            delta = 4e-6f * ((fabsf(ps.o.x) + fabsf(ps.o.y)) + (fabsf(ps.o.z) + sc.bvh_scale));
                        }
                    } else {
                        state = S_READY;
                        best_t = s_t; best_id = s_id;
                    }
                }
            }
            PT_TICK(c4);
            PT_ACC(cyc_trace, c3, c4);
        }
    }
    if (fp.counters) {
        for (int off = 32; off > 0; off >>= 1) { n_bvh_nodes += __shfl_xor(n_bvh_nodes, off); n_bvh_tests += __shfl_xor(n_bvh_tests, off); if (SPH) n_sph_tests += __shfl_xor(n_sph_tests, off); }
        if (lane == 0) {
            if (SPH) atomicAdd(&fp.counters[16], (unsigned long long)n_sph_tests);
            atomicAdd(&fp.counters[17], (kBvhWide ? 4ull : 2ull) * n_bvh_nodes);
            atomicAdd(&fp.counters[0], (unsigned long long)n_samples);
            atomicAdd(&fp.counters[1], (unsigned long long)n_seg);
            atomicAdd(&fp.counters[2], (unsigned long long)n_mesh_seg);
            atomicAdd(&fp.counters[4], (unsigned long long)n_hits);
            atomicAdd(&fp.counters[5], (unsigned long long)n_iter);
            atomicAdd(&fp.counters[6], (unsigned long long)n_prim_mesh_seg);
            atomicAdd(&fp.counters[10], (unsigned long long)n_bvh_nodes);
            atomicAdd(&fp.counters[11], (unsigned long long)n_bvh_tests);
            if (PT_PHASE_TIMERS) { atomicAdd(&fp.counters[12], cyc_gen); atomicAdd(&fp.counters[13], cyc_trace); atomicAdd(&fp.counters[14], cyc_shade); atomicAdd(&fp.counters[15], cyc_mesh); }
            const unsigned long long span = wall_clock64() - t_start;
            atomicAdd(&fp.counters[7], 1ull);
            atomicAdd(&fp.counters[8], span);
            atomicMax(&fp.counters[9], span);
        }
    }
}

// VGPR-resident triangles: persistent single-wave workgroups, 3 waves/SIMD (any mesh size).
template <int SLOTS>
__global__ void __launch_bounds__(kWave, 3) pt_trace_items(SceneView sc, FrameParams fp)
{
    RegStore<SLOTS> st;
    st.init(sc, threadIdx.x);
    trace_items_body<SLOTS>(sc, fp, st, threadIdx.x);
}

// LDS-resident triangles (meshes of one chunk): persistent 4-wave workgroups that share one 25 KB LDS image
// of the chunk; after the staging barrier the waves never synchronise again (each pulls its own items).
constexpr int kLdsBlock = 256;
template <int SLOTS, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(kLdsBlock, WAVES_PER_SIMD) pt_trace_items_lds(SceneView sc, FrameParams fp)
{
    __shared__ float4 ldsA[SLOTS * kWave];
    __shared__ float4 ldsB[SLOTS * kWave];
    __shared__ float ldsC[SLOTS * kWave];
    for (int i = threadIdx.x; i < SLOTS * kWave; i += kLdsBlock) {     // i = slot*64 + lane, the order of the global SoA image
        const float *g = sc.mesh_soa + i;
        const int n = SLOTS * kWave;
        ldsA[i] = make_float4(g[0], g[n], g[2 * n], g[3 * n]);
        ldsB[i] = make_float4(g[4 * n], g[5 * n], g[6 * n], g[7 * n]);
        ldsC[i] = g[8 * n];
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    LdsStore<SLOTS> st;
    st.A = ldsA + lane; st.B = ldsB + lane; st.C = ldsC + lane;
    trace_items_body<SLOTS>(sc, fp, st, lane);
}

// Scheduling pre-pass (no effect on results: a sample is a pure function of (pixel, pass)).  Sort key of a tile =
// 65 x (pixels whose unjittered primary ray passes the root-AABB gate, i.e. brute-force mesh work on the first
// segment and probably on later ones) + (other pixels that hit a scene primitive and so bounce at least once).
constexpr int kMaxTileWeight = 65 * kWave;
__global__ void __launch_bounds__(kWave) pt_tile_weights(SceneView sc, FrameParams fp)
{
    const int lane = threadIdx.x, tile = blockIdx.x;
    int x, y;
    bool gate = false, other = false;
    if (tile_pixel(fp, tile, lane, x, y)) {
        const float W = (float)fp.width, H = (float)fp.height;
        const float X = -(((((float)x + sc.sh.pix_off) / W - 0.5f) * W) / H), Y = ((float)y + sc.sh.pix_off) / H - 0.5f;
        const float *m = fp.iv;
        const v3 d = normalize(V((m[0] * X + m[6]) + m[3] * Y, (m[1] * X + m[7]) + m[4] * Y, (m[2] * X + m[8]) + m[5] * Y));
        const v3 o = V(fp.cam[0], fp.cam[1], fp.cam[2]);
        if (sc.n_mesh > 0 && (sc.sh.flags & kShTraceMesh)) gate = hit_aabb(V(sc.bmin[0], sc.bmin[1], sc.bmin[2]), V(sc.bmax[0], sc.bmax[1], sc.bmax[2]), o, d);
        for (int i = 0; i < sc.n_tris && !other; i++) {
            const float4 p1 = sc.tri_p1[i], e1 = sc.tri_e1[i], e2 = sc.tri_e2[i];
            float t, det;
            other = hit_triangle(V(p1.x, p1.y, p1.z), V(e1.x, e1.y, e1.z), V(e2.x, e2.y, e2.z), o, d, t, det, sc.sh.hit_limit);
        }
        for (int i = 0; i < sc.n_sph && !other; i++) {
            const float4 sp = sc.sph[i];
            float t; bool front;
            other = hit_sphere(V(sp.x, sp.y, sp.z), sp.w, o, d, t, front, sc.sh.hit_limit, (sc.sh.flags & kShSumXyz) != 0);
        }
    }
    const unsigned w = 65u * __popcll(wave_ballot(gate)) + __popcll(wave_ballot(other && !gate));
    if (lane == 0) fp.weights[tile] = w;
}

// Counting sort of the tiles by descending weight: one workgroup, histogram and cursors in LDS.  The order inside
// a weight class is whatever the atomics give; nothing depends on it.
__global__ void __launch_bounds__(1024) pt_sort_tiles(const unsigned *weights, unsigned *order, int *class_count, int n_tiles)
{
    constexpr int kPer = (kMaxTileWeight + 1 + 1023) / 1024;          // bins per thread, in descending key order
    __shared__ unsigned hist[kMaxTileWeight + 1];
    __shared__ unsigned part[1024];
    __shared__ unsigned n_class[2];
    const int tid = threadIdx.x;
    for (int i = tid; i <= kMaxTileWeight; i += 1024) hist[i] = 0;
    if (tid < 2) n_class[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n_tiles; i += 1024) {
        const unsigned w = min(weights[i], (unsigned)kMaxTileWeight);
        atomicAdd(&hist[w], 1u);
        if (w >= 65u) atomicAdd(&n_class[0], 1u);                      // keys >= 65: a mesh-gate pixel
        else if (w >= 1u) atomicAdd(&n_class[1], 1u);                  // keys 1..64: scene primitives only
    }
    __syncthreads();
    // exclusive prefix over the bins in descending key order: per-thread runs of kPer bins, then a block scan
    unsigned local = 0;
    for (int j = 0; j < kPer; j++) { const int b = kMaxTileWeight - (tid * kPer + j); if (b >= 0) local += hist[b]; }
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned run = part[tid] - local;
    for (int j = 0; j < kPer; j++) {
        const int b = kMaxTileWeight - (tid * kPer + j);
        if (b >= 0) { const unsigned n = hist[b]; hist[b] = run; run += n; }
    }
    __syncthreads();
    for (int i = tid; i < n_tiles; i += 1024) order[atomicAdd(&hist[min(weights[i], (unsigned)kMaxTileWeight)], 1u)] = (unsigned)i;
    if (tid == 0) { class_count[0] = (int)n_class[0]; class_count[1] = (int)n_class[1]; class_count[2] = n_tiles - (int)n_class[0] - (int)n_class[1]; }
}

// BVH traversal variant (any mesh size; the default for meshes of more than one register chunk): persistent 4-wave
// workgroups, one ray per lane, per-lane traversal stack in LDS.
template <bool SPH, bool SHORT, bool GEN>
__global__ void __launch_bounds__(kLdsBlock, 4) pt_trace_items_bvh(SceneView sc, FrameParams fp, int stack_entries)
{
    extern __shared__ float4 bvh_lds_small[];    // wave-uniform operands (kSceneLdsVecs float4) | stacks: stack_entries x 4 B x 256 lanes
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    BvhStore st;
    if (!GEN) { sc.sh.hit_limit = kHitLimit; sc.sh.box_slack = kBoxSlack; sc.sh.flags = kShDefault; }     // see trace_items_body_bvh
    const bool staged = stage_scene_lds(bvh_lds_small, sc, fp, threadIdx.x, kLdsBlock);
    __syncthreads();
    st.shade = (lds_cf4v *)bvh_lds_small;
    st.scene = staged ? (lds_cf4v *)(bvh_lds_small + kShadeLdsVecs) : nullptr;
    st.stack = reinterpret_cast<unsigned *>(bvh_lds_small + kSceneLdsVecs) + wave * (stack_entries * kWave) + lane;
    st.nodes = sc.bvh_nodes; st.nodes_q = sc.bvh_nodes_q; st.tris = sc.bvh_tris; st.sph = sc.bvh_sph;
    trace_items_body_bvh<SPH, SHORT, GEN>(sc, fp, st, lane);
}

// The same with the whole BVH staged into LDS once per workgroup: persistent 8-wave workgroups, two per CU; dynamic
// LDS = nodes (112 B each: pt_bvh_format.hpp kBvhLdsNodeVecs) | triangles (48 B each) | spheres (32 B each) | wave-uniform operands (kSceneLdsVecs x 16 B) | stacks
// (stack_entries x 4 B x 512 lanes).
// BLOCK = 512: two workgroups per CU, each with its own copy (trees + stacks up to 80 KB); BLOCK = 1024: ONE workgroup per
// CU whose sixteen waves share one copy, for trees that do not fit twice (up to 160 KB with the stacks) -- the same four
// waves per SIMD either way.
constexpr int kBvhLdsBlock = 512, kBvhLdsBlockWide = 1024;
template <bool SPH, bool SHORT, bool GEN, int BLOCK>
__global__ void __launch_bounds__(BLOCK, 4) pt_trace_items_bvh_lds(SceneView sc, FrameParams fp, int n_nodes, int stack_entries)
{
    constexpr int kBvhLdsBlock = BLOCK;
    extern __shared__ float4 bvh_lds[];
    const int n_sph_rec = SPH && sc.sph_root >= 0 ? 2 * sc.n_sph : 0;
    float4 *nodes = bvh_lds, *tris = bvh_lds + kBvhLdsNodeVecs * n_nodes, *sph = tris + 3 * sc.n_bvh_tris, *scene = sph + n_sph_rec;
    unsigned *stack = reinterpret_cast<unsigned *>(scene + kSceneLdsVecs);
    for (int i = threadIdx.x; i < kBvhLdsNodeVecs * n_nodes; i += kBvhLdsBlock) nodes[i] = sc.bvh_nodes[(i / kBvhLdsNodeVecs) * kBvhNodeVecs + i % kBvhLdsNodeVecs];
    for (int i = threadIdx.x; i < 3 * sc.n_bvh_tris; i += kBvhLdsBlock) tris[i] = sc.bvh_tris[i + i / 3];      // 3 of the 4 vectors of a record
    for (int i = threadIdx.x; i < n_sph_rec; i += kBvhLdsBlock) sph[i] = sc.bvh_sph[i];
    if (!GEN) { sc.sh.hit_limit = kHitLimit; sc.sh.box_slack = kBoxSlack; sc.sh.flags = kShDefault; }     // see trace_items_body_bvh
    const bool staged = stage_scene_lds(scene, sc, fp, threadIdx.x, kBvhLdsBlock);
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    BvhLdsStore st;
    st.shade = (lds_cf4v *)scene;
    st.scene = staged ? (lds_cf4v *)(scene + kShadeLdsVecs) : nullptr;
    st.stack = stack + wave * (stack_entries * kWave) + lane;
    st.nodes = (lds_cf4v *)nodes; st.tris = (lds_cf4v *)tris; st.sph = (lds_cf4v *)sph;
    trace_items_body_bvh<SPH, SHORT, GEN>(sc, fp, st, lane);
}

// rt.comp:337-339 over the passes of one batch, in pass order: accum = mix(accum, radiance(pass), 1/pass)
// (sum_mode: plain sum for the multi-GPU sample split).  One thread per traced pixel.
__global__ void __launch_bounds__(256) pt_accumulate(FrameParams fp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= fp.traced_width) return;
    const size_t plane = (size_t)fp.width * fp.height, pix = (size_t)y * fp.width + x;
    float4 *px = reinterpret_cast<float4 *>(fp.accum) + pix;
    const float4 prev = *px;
    v3 acc = V(prev.x, prev.y, prev.z);
    const Rgb *rad = fp.radiance + pix;
    auto apply = [&](const Rgb c, int k) {
        if (fp.sum_mode) {
            acc = acc + V(c.r, c.g, c.b);
        } else {
            const float t = 1.0f / (float)(fp.first_pass + k);          // rt.comp:337
            acc = mix3(acc, V(c.r, c.g, c.b), t);
        }
    };
    int k = 0;
#ifndef PT_ACC_UNROLL
#define PT_ACC_UNROLL 8
#endif
#ifndef PT_ACC_NT
#define PT_ACC_NT 1          // streaming loads: the workspace is read once (0.35 -> 0.30 ms at 1080p x 64)
#endif
    for (; k + PT_ACC_UNROLL <= fp.n_passes; k += PT_ACC_UNROLL) {      // independent loads in flight per thread
        Rgb c[PT_ACC_UNROLL];
#pragma unroll
        for (int j = 0; j < PT_ACC_UNROLL; j++) {
            const Rgb *q = rad + (size_t)(k + j) * plane;
#if PT_ACC_NT
            c[j] = Rgb{__builtin_nontemporal_load(&q->r), __builtin_nontemporal_load(&q->g), __builtin_nontemporal_load(&q->b)};
#else
            c[j] = *q;
#endif
        }
#pragma unroll
        for (int j = 0; j < PT_ACC_UNROLL; j++) apply(c[j], k + j);
    }
    for (; k < fp.n_passes; k++) apply(rad[(size_t)k * plane], k);
    *px = make_float4(acc.x, acc.y, acc.z, fp.sum_mode ? prev.w + (float)fp.n_passes : 1.0f);
}

// Test-only probe: primary ray + CalculateRay per pixel (pt_read_primary_hits).
template <int SLOTS, class Store>
__device__ __forceinline__ void primary_hits_body(const SceneView &sc, const FrameParams &fp, Store &st, int lane, int tile)
{
    int x, y;
    const bool valid = tile_pixel(fp, tile, lane, x, y);
    v3 o = V(0, 0, 0), d = V(0, 0, 1);
    uint32_t state = 0;
    if (valid) state = primary_ray(fp, sc.sh, x, y, fp.first_pass, o, d);
    unsigned a = 0, b = 0, c = 0, e = 0, f = 0;
    unsigned long long g = 0;
    Hit h = calculate_ray<SLOTS>(sc, st, lane, valid, true, o, d, a, b, c, e, f, g);
    if (!valid) return;
    const size_t i = (size_t)y * fp.width + x;
    if (fp.out_ids) fp.out_ids[i] = h.id;
    if (fp.out_t) fp.out_t[i] = h.t;
    if (fp.out_dirs) { fp.out_dirs[3 * i] = d.x; fp.out_dirs[3 * i + 1] = d.y; fp.out_dirs[3 * i + 2] = d.z; }
    if (fp.out_rng) fp.out_rng[i] = state;
}

template <int SLOTS>
__global__ void __launch_bounds__(kWave, 3) pt_primary_hits(SceneView sc, FrameParams fp)
{
    RegStore<SLOTS> st;
    st.init(sc, threadIdx.x);
    primary_hits_body<SLOTS>(sc, fp, st, threadIdx.x, blockIdx.x);
}

template <int SLOTS>
__global__ void __launch_bounds__(kWave) pt_primary_hits_bvh(SceneView sc, FrameParams fp)
{
    __shared__ unsigned stack[kBvhStack * kWave];
    BvhStore st;
    st.stack = stack + threadIdx.x;
    st.nodes = sc.bvh_nodes; st.nodes_q = sc.bvh_nodes_q; st.tris = sc.bvh_tris; st.sph = sc.bvh_sph;
    primary_hits_body<SLOTS>(sc, fp, st, threadIdx.x, blockIdx.x);
}

// pt_set_option("verify"): pixels whose 16 bytes differ between two RGBA32F images (bitwise, so NaNs compare like any value)
__global__ void pt_count_diff(const uint4 *a, const uint4 *b, size_t n, unsigned *count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool diff = false;
    if (i < n) { const uint4 x = a[i], y = b[i]; diff = (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w); }
    const unsigned long long m = wave_ballot(diff);
    if (m && (threadIdx.x & (kWave - 1)) == 0) atomicAdd(count, (unsigned)__popcll(m));
}

__global__ void pt_scale_kernel(float4 *accum, size_t n, float scale, int set_alpha_one)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = accum[i];
    v.x = v.x * scale; v.y = v.y * scale; v.z = v.z * scale;
    if (set_alpha_one) v.w = 1.0f;
    accum[i] = v;
}

}  // namespace pt
