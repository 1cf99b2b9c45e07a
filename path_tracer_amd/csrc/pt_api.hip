// pt_api.hip — C ABI (include/pt_hip.h) over the HIP kernels in pt_kernels.hpp.
// Host side of the boundary: converts the reference's AoS records (src/Object.h:8-75) into the
// device SoA/AoS images the kernels read, validates every index the kernels will dereference, and
// launches on the context's stream.  No CPU rendering path exists here.
#include "pt_kernels.hpp"
#include "pt_bvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <dlfcn.h>

#pragma clang fp contract(off)

#ifndef PT_DEFAULT_LDS_VARIANT
#define PT_DEFAULT_LDS_VARIANT 4
#endif
#ifndef PT_DEFAULT_VARIANT
#define PT_DEFAULT_VARIANT 6
#endif

using namespace pt;

// ---- pt_render_static's scratch: one array per GPU, in the code object itself -------------------------------------------
// Everything pt_render keeps in buffers of its own beyond the caller's inputs and outputs -- the work-item queue, the tile
// schedule (sort keys, order, class sizes) and the per-pass radiance workspace -- laid out in ONE file-scope device array, so
// that pt_render_static never allocates.  Every GPU the library runs on holds its own copy (at its own address) until the
// process ends.  In host code the array's NAME is only a stand-in: the device address comes from hipGetSymbolAddress, once
// per device (scratch_base below, called by pt_create with that device current), and only that pointer is ever used.
constexpr size_t kScratchBytes = (size_t)64 << 20;             // 64 MiB
constexpr int kScratchMaxTiles = 1 << 16;                      // tiles of 64 px: 1920x1080 has 32,400, 2560x1440 57,600
constexpr size_t kScratchQueueOff = 0, kScratchClassOff = 256, kScratchWeightsOff = 4096,
                 kScratchOrderOff = kScratchWeightsOff + (size_t)kScratchMaxTiles * 4,
                 kScratchRadianceOff = kScratchOrderOff + (size_t)kScratchMaxTiles * 4;       // 528,384: a multiple of 256
static_assert(kScratchRadianceOff % 256 == 0 && kScratchRadianceOff < kScratchBytes, "scratch layout");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

thread_local std::string g_create_error;

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {nullptr};
// One event per device, recorded after the last launch of every pt_render_static call and waited for by the next one
// before it touches the array: contexts launch on streams of their own, and without it a second context's schedule
// kernels would overwrite what the first context's trace kernel is still reading.  Created with the array's address, at
// pt_create time, so that the render path itself creates nothing.
hipEvent_t g_scratch_event[kMaxDevices] = {nullptr};

// device address of `scratch` on `device` (which must be the current device); resolved once per device
hipError_t scratch_base(int device, unsigned char **out)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        unsigned char *p = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        if (!g_scratch_event[device] && (e = hipEventCreateWithFlags(&g_scratch_event[device], hipEventDisableTiming)) != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    hipError_t upload(const void *src, size_t n, hipStream_t s)
    {
        if (n > bytes || !p) {
            if (p) { hipError_t e0 = hipFree(p); if (e0 != hipSuccess) return e0; p = nullptr; bytes = 0; }
            size_t cap = n ? n : 16;
            hipError_t e = hipMalloc(&p, cap);
            if (e != hipSuccess) return e;
            bytes = cap;
        }
        if (n) return hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, s);
        return hipSuccess;
    }
    hipError_t ensure(size_t n)          // at least n bytes, contents undefined
    {
        if (p && n <= bytes) return hipSuccess;
        if (p) { hipError_t e0 = hipFree(p); if (e0 != hipSuccess) return e0; p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e == hipSuccess) bytes = n ? n : 16;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

struct pt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<hipEvent_t> ev;        // pairs (before, after) around every pt_trace_items launch of the last render
    int n_timed = 0;
    const char *last_kernel = "";      // symbol of the trace kernel the last render launched
    unsigned char *scratch_dev = nullptr;   // this device's copy of the static scratch array (pt_render_static)
    DevBuf radiance;                   // [batch passes][H][W] Rgb workspace (12 B per sample)
    size_t workspace_limit = (size_t)4 << 30;
    int chunk_passes = 4;              // passes per work item (mesh tiles; others take multiples)
    DevBuf queue;                      // work-item counter of pt_trace_items
    DevBuf tile_weights, tile_order, class_count;   // schedule of pt_trace_items: tiles sorted heaviest first
    bool order_dirty = true;           // camera, scene, size or stream changed since the schedule was made
    int n_cus = 256;
    int waves_per_cu = 12;             // resident single-wave workgroups per CU at 3 waves/SIMD
    int variant = 0;                   // 0 auto, 1 VGPR-resident triangles, 2/3/4 LDS-resident at 4/5/6 waves per SIMD
    std::string err;

    int width = 0, height = 0;
    float *accum_own = nullptr; size_t accum_own_bytes = 0;
    float *accum = nullptr;            // own or external
    bool accum_external = false;

    // BVHs (pt_bvh.hpp): one tree over the mesh triangles (built by pt_upload_mesh), one over the spheres (pt_upload_scene);
    // their nodes are concatenated (mesh tree first, root 0) and uploaded by sync_bvh before the next launch
    DevBuf bvh_nodes, bvh_nodes_q, bvh_tris, bvh_sph;
    std::vector<BvhNode> h_mesh_nodes, h_sph_nodes;
    bool mesh_bvh_ok = false, sph_bvh_ok = false, nodes_dirty = true;
    float mesh_scale = 0.0f, sph_scale = 0.0f; int mesh_depth = 0, sph_depth = 0;
    bool bvh_ok = false;               // the traversal variants can run (something to traverse, nothing the builder refused in the mesh)
    float bvh_scale = 0.0f; int bvh_depth = 0, bvh_n_nodes = 0, sph_root = -1;
    DevBuf mesh_soa, mesh_p1, mesh_e1, mesh_e2, mesh_nrm, tri_p1, tri_e1, tri_e2, tri_nrm, sph, sph_mat, mats, mat_fres, prim_tab, counters;
    int n_mesh = 0, n_chunks = 0, n_tris = 0, n_sph = 0, n_mats = 0;
    int n_bvh_tris = 0, n_list = 0, n_never = 0;       // mesh triangles in the tree / on the list / that can never be hit (pt_bvh.hpp classify_triangle)
    float sph_rmin = 0.0f, sph_reach = 0.0f;           // smallest |radius| and largest |centre|_1 of the spheres
    std::vector<float4> h_mesh_rec, h_tri_rec;   // winner records (p1 | material, e1, e2, normal) of the mesh / the scene triangles
    bool prim_dirty = true;                      // prim_tab = h_mesh_rec ++ h_tri_rec must be rebuilt
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    float sph_bmin[3] = {0, 0, 0}, sph_bmax[3] = {0, 0, 0};   // bounds of all spheres (pt_upload_scene)
    bool have_mesh = false, have_scene = false, have_camera = false;
    uint32_t max_mat_mesh = 0, max_mat_scene = 0;   // largest material index referenced

    void *group_comm = nullptr;         // ncclComm_t of pt_group_init
    int group_rank = 0, group_world = 0;
    int *group_agree = nullptr;         // device, kGroupAgree ints: the arguments every rank must share (pt_group_render)
    float cam[3] = {0, 0, 0}, iv[9] = {0};
    int max_bounces = 3;              // rt.comp:23
    int debug_normal = 0;
    PtShadingParams shading;          // pt_set_shading_params; pt_create fills in preset 0 (rt.comp)
    std::vector<float> h_ior;         // material.ior of the uploaded materials (the Fresnel table depends on shading.fresnel_n1_one)
    // scheduling options (pt_set_option; 0 = the built-in choice for the launch at hand)
    // pt_set_option("verify", n): every n-th render of the traversal kernels is repeated by the brute-force kernel and compared
    int opt_verify = 0, opt_verify_inject = 0;
    unsigned verify_calls = 0, verify_runs = 0;
    DevBuf verify_before, verify_exact, verify_diff;
    int opt_park_high = 0, opt_park_wait = 0, opt_refill_min = 0, opt_park_exit = 0, opt_ppi1_mul = 4, opt_ppi2_mul = 8, opt_no_fused = 0, opt_wgs_per_cu = 0, opt_lds_block = 0;
};

namespace {

int fail(pt_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                                       \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, PT_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

struct Bind { pt_ctx *c; hipError_t e; explicit Bind(pt_ctx *c_) : c(c_), e(hipSetDevice(c_->device)) {} };

inline float3 f3(const float *p) { return make_float3(p[0], p[1], p[2]); }

// host restatement of the per-triangle precomputation; float32, one rounding per op (this TU is
// compiled with -ffp-contract=off for host and device alike)
void tri_precompute(const PtTriangle &t, float4 &p1, float4 &e1, float4 &e2, float4 &nr)
{
    float e1x = t.p2[0] - t.p1[0], e1y = t.p2[1] - t.p1[1], e1z = t.p2[2] - t.p1[2];
    float e2x = t.p3[0] - t.p1[0], e2y = t.p3[1] - t.p1[1], e2z = t.p3[2] - t.p1[2];
    // cross(e2, e1), then normalize(v) = v * (1/sqrt(dot(v,v))), dot summed z,y,x  (rt.comp:171)
    float cx = e2y * e1z - e2z * e1y, cy = e2z * e1x - e2x * e1z, cz = e2x * e1y - e2y * e1x;
    float dd = (cz * cz + cy * cy) + cx * cx;
    float r = 1.0f / sqrtf(dd);
    uint32_t mi = t.materialIndex; float mf; std::memcpy(&mf, &mi, 4);
    p1 = make_float4(t.p1[0], t.p1[1], t.p1[2], mf);
    e1 = make_float4(e1x, e1y, e1z, 0.0f);
    e2 = make_float4(e2x, e2y, e2z, 0.0f);
    nr = make_float4(cx * r, cy * r, cz * r, 0.0f);
}

int check_ready(pt_ctx *c)
{
    if (!c) return PT_E_INVALID;
    if (c->width < 8 || c->height < 8 || !c->accum) return fail(c, PT_E_STATE, "image size not set (pt_resize)");
    if (!c->have_scene) return fail(c, PT_E_STATE, "scene not uploaded (pt_upload_scene)");
    if (!c->have_mesh) return fail(c, PT_E_STATE, "mesh not uploaded (pt_upload_mesh; n_tris = 0 is allowed)");
    if (!c->have_camera) return fail(c, PT_E_STATE, "camera not set (pt_set_camera)");
    if (c->n_mesh > 0 && c->max_mat_mesh >= (uint32_t)c->n_mats)
        return fail(c, PT_E_INVALID, "mesh references material %u but only %d materials are uploaded", c->max_mat_mesh, c->n_mats);
    if ((c->n_tris > 0 || c->n_sph > 0) && c->max_mat_scene >= (uint32_t)c->n_mats)
        return fail(c, PT_E_INVALID, "scene references material %u but only %d materials are uploaded", c->max_mat_scene, c->n_mats);
    return PT_OK;
}

// PtShadingParams -> what the kernels take (pt_kernels.hpp Shading)
Shading shading_of(const PtShadingParams &p)
{
    Shading h;
    h.hit_limit = p.hit_limit; h.origin_offset = p.origin_offset; h.jitter_scale = p.jitter_scale;
    for (int k = 0; k < 3; k++) {
        volatile float e = p.sky_color[k] * p.sky_intensity;      // one float32 rounding, as the shader's compiler folds the constants
        h.sky[k] = p.sky_color[k]; h.sky_emit[k] = e;
    }
    // a sphere counts as hit down to t1 = -hit_limit (rt.comp:146): boxes that end before that, less a margin, are skipped
    h.box_slack = std::min(kBoxSlack, -10.0f * p.hit_limit);
    h.pix_off = p.pixel_center ? 0.5f : 0.0f;
    h.flags = (p.fresnel_n1_one ? kShN1One : 0) | (p.two_hemisphere_draws ? kShTwoDraws : 0) | (p.sqrt_output ? kShSqrtOut : 0)
              | (p.sphere_sum_xyz ? kShSumXyz : 0) | (p.trace_mesh ? kShTraceMesh : 0);
    return h;
}

// the traversal kernels' GEN instantiation (pt_kernels.hpp trace_items_body_bvh) is needed
bool generic_shading(const Shading &h) { return h.flags != kShDefault || h.hit_limit != kHitLimit || h.box_slack != kBoxSlack; }

// What FresnelReflectAmount (rt.comp:235-254) divides, per material: (1 / ior, r0 of a front-face hit, n1 / n2 of a front-face
// hit, r0 of a back-face hit); a back-face hit takes n1 / n2 from the first entry when it needs one.  Host floats are IEEE
// binary32 and this file is built without contraction, so these are the values the kernel's own divisions would produce.
//   rt.comp:283  n1 = ior = front ? 1 / material.ior : material.ior, n2 = material.ior (a back-face hit has n1 = n2: r0 = 0)
//   rt.frag:283  n1 = 1.0 on both faces: r0 = ((1 - n2) / (1 + n2))^2, n1 / n2 = 1 / n2
int upload_fresnel_table(pt_ctx *c)
{
    const int n = (int)c->h_ior.size();
    std::vector<float> fres((size_t)std::max(1, n) * 4, 0.0f);
    for (int i = 0; i < n; i++) {
        volatile float n2 = c->h_ior[i];
        volatile float inv = 1.0f / n2;                            // rt.comp:281
        if (c->shading.fresnel_n1_one) {
            volatile float q = (1.0f - n2) / (1.0f + n2);
            fres[4 * i + 0] = inv; fres[4 * i + 1] = q * q; fres[4 * i + 2] = inv; fres[4 * i + 3] = q * q;
        } else {
            volatile float q = (inv - n2) / (inv + n2);
            volatile float qb = (n2 - n2) / (n2 + n2);
            fres[4 * i + 0] = inv; fres[4 * i + 1] = q * q; fres[4 * i + 2] = inv / n2; fres[4 * i + 3] = qb * qb;
        }
    }
    HIPCHK(c, c->mat_fres.upload(fres.data(), fres.size() * sizeof(float), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                    // the staging vector dies here
    return PT_OK;
}

SceneView scene_view(const pt_ctx *c)
{
    SceneView s;
    s.sh = shading_of(c->shading);
    s.mesh_soa = (const float *)c->mesh_soa.p;
    s.mesh_p1 = (const float4 *)c->mesh_p1.p; s.mesh_e1 = (const float4 *)c->mesh_e1.p;
    s.mesh_e2 = (const float4 *)c->mesh_e2.p; s.mesh_nrm = (const float4 *)c->mesh_nrm.p;
    s.n_mesh = c->n_mesh; s.n_chunks = c->n_chunks;
    for (int k = 0; k < 3; k++) { s.bmin[k] = c->bmin[k]; s.bmax[k] = c->bmax[k]; }
    s.tri_p1 = (const float4 *)c->tri_p1.p; s.tri_e1 = (const float4 *)c->tri_e1.p;
    s.tri_e2 = (const float4 *)c->tri_e2.p; s.tri_nrm = (const float4 *)c->tri_nrm.p;
    s.n_tris = c->n_tris;
    s.sph = (const float4 *)c->sph.p; s.sph_mat = (const uint32_t *)c->sph_mat.p; s.n_sph = c->n_sph;
    s.mats = (const PtMaterial *)c->mats.p; s.mat_fres = (const float4 *)c->mat_fres.p; s.prim_tab = (const float4 *)c->prim_tab.p;
    s.bvh_nodes = (const float4 *)c->bvh_nodes.p; s.bvh_nodes_q = (const float4 *)c->bvh_nodes_q.p; s.bvh_tris = (const float4 *)c->bvh_tris.p; s.bvh_scale = c->bvh_scale;
    s.bvh_sph = (const float4 *)c->bvh_sph.p; s.sph_root = c->sph_root;
    for (int k = 0; k < 3; k++) { s.sph_bmin[k] = c->sph_bmin[k]; s.sph_bmax[k] = c->sph_bmax[k]; }
    s.n_bvh_tris = c->mesh_bvh_ok ? c->n_bvh_tris : 0;
    s.sph_k = kSphereReachK; s.sph_rmin = c->sph_rmin; s.sph_reach = c->sph_reach;
    return s;
}

// tiles of 64 pixels in the traced area (pt_kernels.hpp tile_pixel)
int tile_count(int width, int height)
{
    const int tw = (width / 8) * 8, rows = (height / 8) * 8;
    return PT_TILE_8X8 ? (tw / 8) * (rows / 8) : ((tw + kWave - 1) / kWave) * rows;
}

FrameParams frame_params(const pt_ctx *c, int first_pass, int n_passes, int sum_mode)
{
    FrameParams f;
    std::memset(&f, 0, sizeof f);
    f.width = c->width; f.height = c->height;
    for (int k = 0; k < 3; k++) f.cam[k] = c->cam[k];
    for (int k = 0; k < 9; k++) f.iv[k] = c->iv[k];
    f.max_bounces = c->max_bounces; f.debug_normal = c->debug_normal;
    f.first_pass = first_pass; f.n_passes = n_passes; f.sum_mode = sum_mode;
    f.chunk = c->chunk_passes;
    f.traced_width = (c->width / 8) * 8; f.rows = (c->height / 8) * 8;
    f.x_blocks = PT_TILE_8X8 ? f.traced_width / 8 : (f.traced_width + kWave - 1) / kWave;
    f.accum = c->accum;
    f.radiance = (Rgb *)c->radiance.p;
    f.counters = (unsigned long long *)c->counters.p;
    return f;
}

// The traversal kernels by [scene has a sphere tree][few passes: fused accumulation][kSh* flags other than rt.comp's]
// (template arguments SPH, SHORT, GEN of pt_kernels.hpp), with the names rocprofv3 reports them under.
struct BvhKernel { const void *fn; const char *name; };
#define PT_BVH_ROW(K, S, F) { { (const void *)&K<S, F, false>, #K "<" #S ", " #F ", false>" }, { (const void *)&K<S, F, true>, #K "<" #S ", " #F ", true>" } }
#define PT_BVH_LDS_ROW(S, F, B) { { (const void *)&pt_trace_items_bvh_lds<S, F, false, B>, "pt_trace_items_bvh_lds<" #S ", " #F ", false, " #B ">" }, \
                                  { (const void *)&pt_trace_items_bvh_lds<S, F, true, B>, "pt_trace_items_bvh_lds<" #S ", " #F ", true, " #B ">" } }
// ... and by [one 1024-thread workgroup per CU instead of two of 512] first
const BvhKernel kBvhLdsKernels[2][2][2][2] = { { { PT_BVH_LDS_ROW(false, false, 512), PT_BVH_LDS_ROW(false, true, 512) },
                                                 { PT_BVH_LDS_ROW(true, false, 512), PT_BVH_LDS_ROW(true, true, 512) } },
                                               { { PT_BVH_LDS_ROW(false, false, 1024), PT_BVH_LDS_ROW(false, true, 1024) },
                                                 { PT_BVH_LDS_ROW(true, false, 1024), PT_BVH_LDS_ROW(true, true, 1024) } } };
#undef PT_BVH_LDS_ROW
const BvhKernel kBvhKernels[2][2][2] = { { PT_BVH_ROW(pt_trace_items_bvh, false, false), PT_BVH_ROW(pt_trace_items_bvh, false, true) },
                                         { PT_BVH_ROW(pt_trace_items_bvh, true, false), PT_BVH_ROW(pt_trace_items_bvh, true, true) } };
#undef PT_BVH_ROW

// Per-lane stack entries beyond the trees' depth: the permanent bottom entry, the other tree's root (counted in bvh_depth), and
// the free slot above the stack that every node visit writes unconditionally.
constexpr int kBvhStackExtra = 3;
// LDS the LDS-resident BVH kernel needs per workgroup of `block` threads: two workgroups of 512 must fit in a CU's 160 KB,
// or one of 1024 (a little is left for the runtime).
constexpr size_t kBvhLdsLimit = 80 * 1024, kBvhLdsLimitWide = 158 * 1024;
size_t bvh_lds_bytes(const pt_ctx *c, int block)
{
    return (size_t)c->bvh_n_nodes * (kBvhLdsNodeVecs * 16) + (size_t)c->n_bvh_tris * 48 + (c->sph_root >= 0 ? (size_t)c->n_sph * 32 : 0) + (size_t)kSceneLdsVecs * 16
           + (size_t)(c->bvh_depth + kBvhStackExtra) * 4 * (size_t)block;
}
// The LDS-resident traversal kernels take more dynamic LDS than a kernel may by default: the cap is lifted for every
// instantiation once per device, at pt_create, so that no render call -- pt_render_static under graph capture in particular --
// has to touch function attributes.  The attribute only permits; what a launch takes is its own argument.
std::mutex g_attr_mutex;
bool g_attr_done[kMaxDevices] = {false};
hipError_t allow_large_lds(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    if (g_attr_done[device]) return hipSuccess;
    for (int wide = 0; wide < 2; wide++)
        for (int sph = 0; sph < 2; sph++)
            for (int fused = 0; fused < 2; fused++)
                for (int gen = 0; gen < 2; gen++) {
                    const hipError_t e = hipFuncSetAttribute(kBvhLdsKernels[wide][sph][fused][gen].fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             (int)(wide ? kBvhLdsLimitWide : kBvhLdsLimit));
                    if (e != hipSuccess) return e;
                }
    g_attr_done[device] = true;
    return hipSuccess;
}

// 512 when the trees fit twice per CU, else 1024 (one shared copy per CU), 0: not at all (variant 5 reads them from global memory)
int bvh_lds_block(const pt_ctx *c)
{
    if (c->opt_lds_block == kBvhLdsBlock || c->opt_lds_block == kBvhLdsBlockWide)
        return bvh_lds_bytes(c, c->opt_lds_block) <= (c->opt_lds_block == kBvhLdsBlock ? kBvhLdsLimit : kBvhLdsLimitWide) ? c->opt_lds_block : 0;
    if (bvh_lds_bytes(c, kBvhLdsBlock) <= kBvhLdsLimit) return kBvhLdsBlock;
    if (bvh_lds_bytes(c, kBvhLdsBlockWide) <= kBvhLdsLimitWide) return kBvhLdsBlockWide;
    return 0;
}

// Concatenate the two trees' nodes (mesh first: its root is node 0; the sphere tree's child indices are shifted) and upload.
int sync_bvh(pt_ctx *c)
{
    if (c->prim_dirty) {
        // one table of winner records indexed by canonical id (mesh triangles, then scene triangles): 64 B = one cache
        // line per hit and ONE base pointer in the traversal kernels, where eight separate arrays cost 16 SGPRs
        std::vector<float4> all(c->h_mesh_rec);
        all.insert(all.end(), c->h_tri_rec.begin(), c->h_tri_rec.end());
        HIPCHK(c, c->prim_tab.upload(all.data(), all.size() * sizeof(float4), c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->prim_dirty = false;
    }
    if (!c->nodes_dirty) return PT_OK;
    const bool mesh = c->n_mesh > 0 && c->mesh_bvh_ok, sph = c->n_sph > 0 && c->sph_bvh_ok;
    c->bvh_ok = (c->n_mesh == 0 || c->mesh_bvh_ok) && (mesh || sph);
    c->sph_root = -1; c->bvh_n_nodes = 0; c->bvh_depth = 0; c->bvh_scale = 0.0f;
    if (c->bvh_ok) {
        // stack entries a lane can need below the roots: the binary tree pushes one sibling per level, the four-wide collapse up
        // to three per node (pt_bvh.hpp collapse_bvh4 computes the exact bound)
        int need_mesh = mesh ? c->mesh_depth : 0, need_sph = sph ? c->sph_depth : 0;
        const void *data; size_t bytes;
        std::vector<BvhNode> all;
        std::vector<BvhNode4> wide;
        std::vector<BvhNode4Q> quant;          // staging vectors live until the synchronisation below
        if (kBvhWide) {
            Bvh4 m4, s4;
            if (mesh) { collapse_bvh4(c->h_mesh_nodes, m4, 0); need_mesh = m4.stack_need; }
            if (sph) { c->sph_root = (int)m4.nodes.size(); collapse_bvh4(c->h_sph_nodes, s4, c->sph_root); need_sph = s4.stack_need; }
            wide = std::move(m4.nodes);
            wide.insert(wide.end(), s4.nodes.begin(), s4.nodes.end());
            if (c->opt_verify_inject && mesh && !wide.empty()) {
                // test hook for the "verify" option: a deliberately WRONG tree -- the root's child boxes shrunk to half their
                // size -- so that the traversal skips triangles it must not skip
                BvhNode4 &r0 = wide[0];
                float *lo[3] = {r0.mnx, r0.mny, r0.mnz}, *hi[3] = {r0.mxx, r0.mxy, r0.mxz};
                for (int i = 0; i < 4; i++)
                    for (int k = 0; k < 3; k++) {
                        if (r0.child[i] == kBvhNoChild) continue;
                        const float mid = 0.5f * (lo[k][i] + hi[k][i]), half = 0.25f * (hi[k][i] - lo[k][i]);
                        lo[k][i] = mid - half; hi[k][i] = mid + half;
                    }
            }
            data = wide.data(); bytes = wide.size() * sizeof(BvhNode4);
            c->bvh_n_nodes = (int)wide.size();
            // the 64-byte form, for the kernel that reads the tree in place (variant 5)
            quant.resize(wide.size());
            bool q_ok = true;
            for (size_t i = 0; i < wide.size() && q_ok; i++) q_ok = quantise_bvh4(wide[i], quant[i]);
            if (!q_ok) { need_mesh = need_sph = kBvhMaxDepth; }       // refused below: brute force
            else HIPCHK(c, c->bvh_nodes_q.upload(quant.data(), quant.size() * sizeof(BvhNode4Q), c->stream));
        } else {
            if (mesh) all = c->h_mesh_nodes;
            if (sph) {
                const int base = (int)all.size();
                c->sph_root = base;
                for (BvhNode nd : c->h_sph_nodes) {
                    if (nd.left >= 0) nd.left += base;
                    if (nd.right >= 0) nd.right += base;
                    all.push_back(nd);
                }
            }
            if (c->opt_verify_inject && mesh && !all.empty()) {
                BvhNode &r0 = all[0];
                for (int k = 0; k < 3; k++) {
                    const float lc = 0.5f * (r0.lmin[k] + r0.lmax[k]), lh = 0.25f * (r0.lmax[k] - r0.lmin[k]);
                    const float rc = 0.5f * (r0.rmin[k] + r0.rmax[k]), rh = 0.25f * (r0.rmax[k] - r0.rmin[k]);
                    r0.lmin[k] = lc - lh; r0.lmax[k] = lc + lh; r0.rmin[k] = rc - rh; r0.rmax[k] = rc + rh;
                }
            }
            data = all.data(); bytes = all.size() * sizeof(BvhNode);
            c->bvh_n_nodes = (int)all.size();
        }
        if (std::max(need_mesh, need_sph) + 1 > kBvhMaxDepth) {
            // deeper than the traversal stacks are sized for (the test-only probe kernel's is static): brute force instead
            c->bvh_ok = false; c->sph_root = -1; c->bvh_n_nodes = 0;
        } else {
            HIPCHK(c, c->bvh_nodes.upload(data, bytes, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->bvh_depth = std::max(need_mesh, need_sph) + ((mesh && sph) ? 1 : 0);   // + the other tree's root on the stack, where there are two
            c->bvh_scale = std::max(mesh ? c->mesh_scale : 0.0f, sph ? c->sph_scale : 0.0f);
        }
    }
    c->nodes_dirty = false;
    return PT_OK;
}

// Kernel variant for the uploaded mesh.  Default: the BVH traversal whenever the builder accepted the mesh; brute
// force from the LDS image (one register chunk) or from VGPRs otherwise, or when asked for through pt_set_tuning.
int pick_variant(const pt_ctx *c)
{
    int variant = c->variant ? c->variant : PT_DEFAULT_VARIANT;
    if (variant >= 5 && !c->bvh_ok) variant = PT_DEFAULT_LDS_VARIANT;     // no mesh, or coordinates the builder refused
    if (variant == 6 && bvh_lds_block(c) == 0) variant = 5;               // too large for a CU's LDS

    if (variant >= 2 && variant <= 4 && c->n_chunks != 1) variant = 1;    // the LDS image holds exactly one chunk
    return variant;
}

// no_alloc (pt_render_static): queue, schedule and workspace live in the static scratch array; nothing is allocated, created or
// freed on the way (no hipMalloc / hipFree / hipEventCreate), and sizes the array cannot hold are refused.
int launch_trace_impl(pt_ctx *c, int first_pass, int n_passes, int sum_mode, bool no_alloc);

int launch_trace(pt_ctx *c, int first_pass, int n_passes, int sum_mode, bool no_alloc = false)
{
    if (!no_alloc) return launch_trace_impl(c, first_pass, n_passes, sum_mode, false);
    // pt_render_static: every context on this GPU uses the same array.  The whole call is enqueued under the lock, behind
    // the event the previous call (of whatever context) recorded after its last launch, and records that event again at its
    // end -- two stream operations, nothing allocated.  Under stream capture the event calls are left out (an event
    // recorded outside the capture cannot be waited for inside it): a captured call is ordered by the graph's own edges, and
    // the caller must not run two such graphs, or a graph and a direct call, on one GPU at the same time.
    if (!c) return PT_E_INVALID;
    Bind b(c); HIPCHK(c, b.e);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    HIPCHK(c, hipStreamIsCapturing(c->stream, &cap));
    hipEvent_t ev = (c->device >= 0 && c->device < kMaxDevices) ? g_scratch_event[c->device] : nullptr;
    const bool ordered = cap == hipStreamCaptureStatusNone && ev != nullptr;
    if (ordered) HIPCHK(c, hipStreamWaitEvent(c->stream, ev, 0));
    const int rc = launch_trace_impl(c, first_pass, n_passes, sum_mode, true);
    // recorded even after a failure part-way: whatever was enqueued before it still reads the array
    if (ordered) { const hipError_t e = hipEventRecord(ev, c->stream); if (rc == PT_OK && e != hipSuccess) return fail(c, PT_E_HIP, "hipEventRecord: %s", hipGetErrorString(e)); }
    return rc;
}

int launch_trace_impl(pt_ctx *c, int first_pass, int n_passes, int sum_mode, bool no_alloc)
{
    int rc = check_ready(c);
    if (rc) return rc;
    if (first_pass < 0 || n_passes < 1) return fail(c, PT_E_INVALID, "bad pass range %d+%d", first_pass, n_passes);
    Bind b(c); HIPCHK(c, b.e);
    if (no_alloc && (c->prim_dirty || c->nodes_dirty)) return fail(c, PT_E_STATE, "scene tables not on the device yet (upload the mesh and the scene first)");
    if (!no_alloc && (rc = sync_bvh(c)) != PT_OK) return rc;
    const size_t plane_bytes = (size_t)c->width * c->height * sizeof(Rgb);
    const int n_tiles = tile_count(c->width, c->height);
    if (no_alloc) {
        if (!c->scratch_dev) return fail(c, PT_E_STATE, "the static scratch array was not resolved on this device");
        if (n_tiles > kScratchMaxTiles)
            return fail(c, PT_E_LIMIT, "pt_render_static: %d x %d has %d tiles of 64 pixels, the static scratch holds %d (use pt_render)", c->width, c->height, n_tiles, kScratchMaxTiles);
    }
    // The traversal kernels (variants 5, 6) schedule lane by lane, so one lane can take ALL passes of a pixel and keep the
    // running mean in registers: no per-pass workspace, no accumulate kernel.  Only for short pass ranges (interactive
    // use: one or a few passes per call): a lane task of 64 passes of a hard pixel runs for milliseconds and, started late,
    // becomes the tail of the launch (measured on config 2: 10.6 ms fused vs 7.8 ms with 4-pass items).
    const bool allow_fused = !c->opt_no_fused;
    const int variant0 = pick_variant(c);
    const bool can_fuse = allow_fused && (variant0 == 5 || variant0 == 6);
    // the kernels index the workspace with 32 bits: passes in a batch x pixels < 2^32
    const size_t ws_limit = no_alloc ? kScratchBytes - kScratchRadianceOff : c->workspace_limit;
    const size_t batch_cap = std::min<size_t>(ws_limit / plane_bytes, 0xffffffffull / ((size_t)c->width * c->height));
    bool fused = can_fuse && n_passes <= c->chunk_passes;
    int batch = fused ? n_passes : (int)std::max<size_t>(1, std::min<size_t>((size_t)n_passes, batch_cap));
    if (no_alloc && !fused && batch_cap < (size_t)std::min(n_passes, c->chunk_passes + 1)) {
        // the static workspace holds too few passes of this image to be worth a launch of its own: take the passes in short
        // fused launches instead (bit-identical: the running mean continues in the image), or give up where those cannot run
        if (!can_fuse) {
            if (batch_cap < 1) return fail(c, PT_E_LIMIT, "pt_render_static: one pass of %d x %d needs %zu bytes of workspace, the static scratch has %zu (use pt_render)", c->width, c->height, plane_bytes, ws_limit);
        } else {
            fused = true; batch = c->chunk_passes;
        }
    }
    if (!fused && !no_alloc && c->radiance.bytes < plane_bytes * batch) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->radiance.release();
        HIPCHK(c, hipMalloc(&c->radiance.p, plane_bytes * batch));
        c->radiance.bytes = plane_bytes * batch;
    }
    const int n_batches = (n_passes + batch - 1) / batch;
    const bool timed = !no_alloc;      // pt_render_static records no timing events (they could not be read back after a graph capture)
    while (!no_alloc && (int)c->ev.size() < 2 * n_batches) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->ev.push_back(e); }
    SceneView sv = scene_view(c);
    c->n_timed = 0;
    unsigned *q_ptr, *w_ptr, *o_ptr; int *cc_ptr; Rgb *rad_ptr;
    if (no_alloc) {
        q_ptr = (unsigned *)(c->scratch_dev + kScratchQueueOff); cc_ptr = (int *)(c->scratch_dev + kScratchClassOff);
        w_ptr = (unsigned *)(c->scratch_dev + kScratchWeightsOff); o_ptr = (unsigned *)(c->scratch_dev + kScratchOrderOff);
        rad_ptr = (Rgb *)(c->scratch_dev + kScratchRadianceOff);
        c->order_dirty = true;      // the array is shared by every context on this GPU and keeps the last call's contents
    } else {
        if (c->tile_order.bytes < (size_t)n_tiles * sizeof(unsigned)) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->tile_weights.release(); c->tile_order.release();
            HIPCHK(c, hipMalloc(&c->tile_weights.p, (size_t)n_tiles * sizeof(unsigned))); c->tile_weights.bytes = (size_t)n_tiles * sizeof(unsigned);
            HIPCHK(c, hipMalloc(&c->tile_order.p, (size_t)n_tiles * sizeof(unsigned))); c->tile_order.bytes = (size_t)n_tiles * sizeof(unsigned);
            if (!c->class_count.p) { HIPCHK(c, hipMalloc(&c->class_count.p, 4 * sizeof(int))); c->class_count.bytes = 4 * sizeof(int); }
            c->order_dirty = true;
        }
        q_ptr = (unsigned *)c->queue.p; cc_ptr = (int *)c->class_count.p;
        w_ptr = (unsigned *)c->tile_weights.p; o_ptr = (unsigned *)c->tile_order.p; rad_ptr = (Rgb *)c->radiance.p;
    }
    for (int bi = 0; bi < n_batches; bi++) {
        const int p0 = first_pass + bi * batch, np = std::min(batch, n_passes - bi * batch);
        FrameParams fp = frame_params(c, p0, np, sum_mode);
        fp.radiance = rad_ptr;
        // at most one item per (tile, pass): an upper bound for the grid and the 31-bit item index
        const long long items = (long long)n_tiles * ((np + fp.chunk - 1) / fp.chunk);
        if (items > 0x7fffffffll) return fail(c, PT_E_LIMIT, "too many work items in one batch (%lld); lower the workspace limit", items);
        fp.queue = q_ptr; fp.n_tiles = n_tiles;
        // passes per work item: mesh tiles `chunk`; tiles that only bounce off scene primitives and sky tiles take more
        // passes per item (their items are short, and every item costs one ~11 ns device-wide atomic on the queue)
        const int mul1 = c->opt_ppi1_mul, mul2 = c->opt_ppi2_mul;
        fp.class_ppi[0] = fp.chunk; fp.class_ppi[1] = std::max(1, std::min(np, mul1 * fp.chunk)); fp.class_ppi[2] = std::max(1, std::min(np, mul2 * fp.chunk));
        if (fused) fp.class_ppi[0] = fp.class_ppi[1] = fp.class_ppi[2] = np;   // SHORT kernels: every item covers all passes
        for (int k = 0; k < 3; k++) fp.per_tile[k] = (np + fp.class_ppi[k] - 1) / fp.class_ppi[k];
        // Scheduling of the traversal kernels' lane state machine, tuned on configs 2, 3, 5, 4K and the
        // short launches; pt_set_option overrides them per context for experiments (0 = these defaults).
        //   long launches: traverse at 32 parked lanes or after 4 iterations of waiting, leave the traversal phase with 3/8
        //   of its rays unfinished, refill at 4 idle lanes; short launches (tail-bound) and scenes with a sphere tree
        //   (every ray traverses) keep the earlier, more eager settings where those measured better.
        const int e_high = c->opt_park_high, e_wait = c->opt_park_wait, e_refill = c->opt_refill_min, e_exit = c->opt_park_exit;
        const bool big = variant0 == 5;       // BVH in global memory (large meshes: most rays traverse): larger batches
        fp.park_high = e_high ? e_high : (fused ? 40 : big ? 48 : 32);
        fp.park_wait = e_wait ? e_wait : (fused ? 2 : big ? 6 : 4);
        fp.refill_min = e_refill ? e_refill : (fused ? 8 : 4);
        fp.park_exit = e_exit ? e_exit : ((fused || c->sph_root >= 0) ? 2 : 3);

        fp.weights = w_ptr; fp.order = o_ptr; fp.class_count = cc_ptr;
        if (c->order_dirty) {
            hipLaunchKernelGGL(pt_tile_weights, dim3((unsigned)n_tiles), dim3(kWave), 0, c->stream, sv, fp);
            HIPCHK(c, hipGetLastError());
            hipLaunchKernelGGL(pt_sort_tiles, dim3(1), dim3(1024), 0, c->stream, fp.weights, o_ptr, cc_ptr, n_tiles);
            HIPCHK(c, hipGetLastError());
            c->order_dirty = false;
        }
        const int variant = pick_variant(c);
        HIPCHK(c, hipMemsetAsync(q_ptr, 0, sizeof(unsigned), c->stream));
        if (timed) HIPCHK(c, hipEventRecord(c->ev[2 * bi], c->stream));
        if (variant == 1) {
            const unsigned grid = (unsigned)std::min<long long>(items, (long long)c->n_cus * c->waves_per_cu);
            hipLaunchKernelGGL(pt_trace_items<kSlots>, dim3(grid), dim3(kWave), 0, c->stream, sv, fp);
            c->last_kernel = "pt_trace_items<11>";
        } else if (variant == 2) {
            const unsigned grid = (unsigned)std::min<long long>((items + 3) / 4, (long long)c->n_cus * 4);
            hipLaunchKernelGGL((pt_trace_items_lds<kSlots, 4>), dim3(grid), dim3(kLdsBlock), 0, c->stream, sv, fp);
            c->last_kernel = "pt_trace_items_lds<11, 4>";
        } else if (variant == 3) {
            const unsigned grid = (unsigned)std::min<long long>((items + 3) / 4, (long long)c->n_cus * 5);
            hipLaunchKernelGGL((pt_trace_items_lds<kSlots, 5>), dim3(grid), dim3(kLdsBlock), 0, c->stream, sv, fp);
            c->last_kernel = "pt_trace_items_lds<11, 5>";
        } else if (variant == 6) {
            const int block = bvh_lds_block(c), waves = block / kWave;         // pick_variant made sure it is not 0
            const int per_cu = c->opt_wgs_per_cu ? c->opt_wgs_per_cu : (block == kBvhLdsBlock ? 2 : 1);
            const unsigned grid = (unsigned)std::min<long long>((items + waves - 1) / waves, (long long)c->n_cus * per_cu);
            const size_t lds = bvh_lds_bytes(c, block);
            const BvhKernel &bk = kBvhLdsKernels[block == kBvhLdsBlockWide][c->sph_root >= 0][fused][generic_shading(sv.sh)];
            const void *kern = bk.fn;      // its dynamic-LDS cap was lifted once per device (allow_large_lds, pt_create)
            int n_nodes = c->bvh_n_nodes, entries = c->bvh_depth + kBvhStackExtra;
            void *args[] = {&sv, &fp, &n_nodes, &entries};
            HIPCHK(c, hipLaunchKernel(kern, dim3(grid), dim3(block), args, lds, c->stream));
            c->last_kernel = bk.name;
        } else if (variant == 5) {
            // as many 4-wave workgroups per CU as their stacks leave room for (at most 4: 124 VGPRs)
            const size_t lds = (size_t)kSceneLdsVecs * 16 + (size_t)(c->bvh_depth + kBvhStackExtra) * 4 * kLdsBlock;
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / (lds + 512)));
            const unsigned grid = (unsigned)std::min<long long>((items + 3) / 4, (long long)c->n_cus * per_cu);
            const BvhKernel &bk = kBvhKernels[c->sph_root >= 0][fused][generic_shading(sv.sh)];
            const void *kern = bk.fn;
            int entries = c->bvh_depth + kBvhStackExtra;
            void *args[] = {&sv, &fp, &entries};
            HIPCHK(c, hipLaunchKernel(kern, dim3(grid), dim3(kLdsBlock), args, lds, c->stream));
            c->last_kernel = bk.name;
        } else {
            const unsigned grid = (unsigned)std::min<long long>((items + 3) / 4, (long long)c->n_cus * 6);
            hipLaunchKernelGGL((pt_trace_items_lds<kSlots, 6>), dim3(grid), dim3(kLdsBlock), 0, c->stream, sv, fp);
            c->last_kernel = "pt_trace_items_lds<11, 6>";
        }
        HIPCHK(c, hipGetLastError());
        if (timed) HIPCHK(c, hipEventRecord(c->ev[2 * bi + 1], c->stream));
        if (!fused) {
            hipLaunchKernelGGL(pt_accumulate, dim3((fp.traced_width + 255) / 256, fp.rows), dim3(256), 0, c->stream, fp);
            HIPCHK(c, hipGetLastError());
        }
        if (timed) c->n_timed = bi + 1;
    }
    if (no_alloc) c->order_dirty = true;    // the next pt_render must not trust a schedule that lives in the shared array
    return PT_OK;
}

// pt_render / pt_render_pass / pt_render_sum with the "verify" option: every n-th call that would run a traversal kernel is
// first done by the brute-force kernel (exact by construction: every mesh triangle behind the gate, like rt.comp:213-218)
// from the same starting image, then by the traversal kernel, and the two images are compared bit for bit on the device.
// A difference returns PT_E_STATE and leaves the brute-force image in place.  It turns "the traversal's skipping margins have
// never been seen to lose a hit"  into something a user's own scene can assert, at about 4x the cost of the
// checked call.  Counters count both renders; timing and pt_last_kernel_name are the traversal render's.
int render_checked(pt_ctx *c, int first_pass, int n_passes, int sum_mode)
{
    if (!c) return PT_E_INVALID;
    if (c->opt_verify <= 0) return launch_trace(c, first_pass, n_passes, sum_mode);
    int rc = check_ready(c);
    if (rc) return rc;
    Bind b(c); HIPCHK(c, b.e);
    if ((rc = sync_bvh(c)) != PT_OK) return rc;
    if (pick_variant(c) < 5 || (++c->verify_calls % (unsigned)c->opt_verify) != 0) return launch_trace(c, first_pass, n_passes, sum_mode);
    const size_t n_px = (size_t)c->width * c->height, bytes = n_px * 4 * sizeof(float);
    HIPCHK(c, c->verify_before.ensure(bytes));
    HIPCHK(c, c->verify_exact.ensure(bytes));
    HIPCHK(c, c->verify_diff.ensure(sizeof(unsigned)));
    HIPCHK(c, hipMemcpyAsync(c->verify_before.p, c->accum, bytes, hipMemcpyDeviceToDevice, c->stream));
    const int saved = c->variant;
    c->variant = c->n_chunks == 1 ? PT_DEFAULT_LDS_VARIANT : 1;
    rc = launch_trace(c, first_pass, n_passes, sum_mode);
    c->variant = saved;
    if (rc != PT_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->verify_exact.p, c->accum, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->accum, c->verify_before.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = launch_trace(c, first_pass, n_passes, sum_mode)) != PT_OK) return rc;
    HIPCHK(c, hipMemsetAsync(c->verify_diff.p, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(pt_count_diff, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, c->stream, (const uint4 *)c->accum,
                       (const uint4 *)c->verify_exact.p, n_px, (unsigned *)c->verify_diff.p);
    HIPCHK(c, hipGetLastError());
    unsigned differing = 0;
    HIPCHK(c, hipMemcpyAsync(&differing, c->verify_diff.p, sizeof differing, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->verify_runs++;
    if (differing) {
        HIPCHK(c, hipMemcpyAsync(c->accum, c->verify_exact.p, bytes, hipMemcpyDeviceToDevice, c->stream));
        return fail(c, PT_E_STATE, "verify: %u of %zu pixels differ between the traversal kernel (%s) and the brute-force kernel for passes %d..%d; "
                    "the image holds the brute-force result (pt_set_tuning variant 4 selects that kernel for good)", differing, n_px, c->last_kernel,
                    first_pass, first_pass + n_passes - 1);
    }
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_version(void) { return 1; }
const char *pt_kernel_name(void) { return PT_DEFAULT_VARIANT >= 5 ? "pt_trace_items_bvh_lds" : "pt_trace_items_lds"; }

const char *pt_last_kernel_name(const pt_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

const char *pt_last_error(const pt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pt_create(pt_ctx **out, int device)
{
    if (!out) return fail(nullptr, PT_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, PT_E_HIP, "no HIP device available (%s); this library has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, PT_E_INVALID, "device %d out of range [0,%d)", device, n);
    pt_ctx *c = new pt_ctx();
    c->device = device;
    pt_shading_preset(&c->shading, 0);
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(nullptr, PT_E_HIP, "context setup failed: %s", hipGetErrorString(e));
        delete c;
        return PT_E_HIP;
    }
    c->stream = c->own_stream;
    // this device's copy of the static scratch array (pt_render_static): resolved here, once per device, with it current
    if ((e = scratch_base(device, &c->scratch_dev)) != hipSuccess) {
        fail(nullptr, PT_E_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return PT_E_HIP;
    }
    if ((e = allow_large_lds(device)) != hipSuccess) {
        fail(nullptr, PT_E_HIP, "hipFuncSetAttribute(max dynamic LDS) on device %d: %s", device, hipGetErrorString(e));
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return PT_E_HIP;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    unsigned long long zero[kNumCounters] = {0};
    if ((e = c->queue.upload(zero, 16, c->stream)) != hipSuccess) { fail(nullptr, PT_E_HIP, "queue setup failed: %s", hipGetErrorString(e)); delete c; return PT_E_HIP; }
    if ((e = c->counters.upload(zero, sizeof zero, c->stream)) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
        fail(nullptr, PT_E_HIP, "counter setup failed: %s", hipGetErrorString(e));
        delete c;
        return PT_E_HIP;
    }
    *out = c;
    return PT_OK;
}

void pt_destroy(pt_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->group_comm) (void)pt_group_finalize(c);
    for (DevBuf *b : {&c->mesh_soa, &c->mesh_p1, &c->mesh_e1, &c->mesh_e2, &c->mesh_nrm, &c->tri_p1, &c->tri_e1, &c->tri_e2,
                      &c->tri_nrm, &c->sph, &c->sph_mat, &c->mats, &c->mat_fres, &c->prim_tab, &c->counters, &c->queue, &c->tile_weights, &c->tile_order, &c->class_count, &c->bvh_nodes, &c->bvh_nodes_q, &c->bvh_tris, &c->bvh_sph}) b->release();
    if (c->accum_own) (void)hipFree(c->accum_own);
    if (c->group_agree) (void)hipFree(c->group_agree);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    c->radiance.release();
    c->verify_before.release(); c->verify_exact.release(); c->verify_diff.release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int pt_resize(pt_ctx *c, int width, int height)
{
    if (!c) return PT_E_INVALID;
    if (width < 8 || height < 8) return fail(c, PT_E_INVALID, "image must be at least 8x8 (glDispatchCompute(W/8, H/8, 1))");
    if ((long long)width * height > (1ll << 28)) return fail(c, PT_E_LIMIT, "image larger than 2^28 pixels");
    Bind b(c); HIPCHK(c, b.e);
    size_t bytes = (size_t)width * height * 4 * sizeof(float);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bytes > c->accum_own_bytes) {
        if (c->accum_own) { HIPCHK(c, hipFree(c->accum_own)); c->accum_own = nullptr; c->accum_own_bytes = 0; }
        HIPCHK(c, hipMalloc((void **)&c->accum_own, bytes));
        c->accum_own_bytes = bytes;
    }
    c->width = width; c->height = height; c->order_dirty = true;
    c->accum = c->accum_own; c->accum_external = false;
    HIPCHK(c, hipMemsetAsync(c->accum, 0, bytes, c->stream));
    return PT_OK;
}

int pt_set_accum_buffer(pt_ctx *c, void *dev, size_t bytes)
{
    if (!c) return PT_E_INVALID;
    if (c->width < 8) return fail(c, PT_E_STATE, "call pt_resize first");
    if (!dev) { c->accum = c->accum_own; c->accum_external = false; return PT_OK; }
    size_t need = (size_t)c->width * c->height * 4 * sizeof(float);
    if (bytes < need) return fail(c, PT_E_INVALID, "external accumulation buffer too small: %zu < %zu", bytes, need);
    if (((uintptr_t)dev & 15u) != 0) return fail(c, PT_E_INVALID, "external accumulation buffer must be 16-byte aligned");
    c->accum = (float *)dev; c->accum_external = true;
    return PT_OK;
}

int pt_set_stream(pt_ctx *c, void *s)
{
    if (!c) return PT_E_INVALID;
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    c->order_dirty = true;
    c->n_timed = 0;
    return PT_OK;
}

int pt_upload_mesh(pt_ctx *c, const PtTriangle *tris, int n, const PtNode *nodes, int n_nodes)
{
    if (!c) return PT_E_INVALID;
    if (n < 0 || (n > 0 && !tris)) return fail(c, PT_E_INVALID, "bad mesh triangle array");
    if (n > 0 && (n_nodes < 1 || !nodes)) return fail(c, PT_E_INVALID, "a mesh needs nodes[0] (root AABB, rt.comp:213)");
    if (n > (1 << 24)) return fail(c, PT_E_LIMIT, "more than 2^24 mesh triangles");
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int chunks = (n + kChunk - 1) / kChunk;
    std::vector<float4> p1(n), e1(n), e2(n), nr(n);
    // Padded slots of the last chunk hold copies of the LAST real triangle: a copy computes the same t as the
    // original and sits in a later slot / higher lane, so it loses every tie and never changes a result (and,
    // unlike a degenerate filler, it keeps the fast reciprocal on its exact path).
    std::vector<float> soa((size_t)chunks * 9 * kChunk, 0.0f);
    uint32_t maxm = 0;
    for (int i = 0; i < n; i++) {
        tri_precompute(tris[i], p1[i], e1[i], e2[i], nr[i]);
        if (tris[i].materialIndex > maxm) maxm = tris[i].materialIndex;
    }
    for (int j = 0; j < chunks * kChunk; j++) {
        const int i = std::min(j, n - 1);
        const int chunk = j / kChunk, r = j % kChunk, lane = r / kSlots, slot = r % kSlots;
        float *base = soa.data() + (size_t)chunk * 9 * kChunk + slot * kWave + lane;
        const float comp[9] = {p1[i].x, p1[i].y, p1[i].z, e1[i].x, e1[i].y, e1[i].z, e2[i].x, e2[i].y, e2[i].z};
        for (int k = 0; k < 9; k++) base[(size_t)k * kChunk] = comp[k];
    }
    HIPCHK(c, c->mesh_soa.upload(soa.data(), soa.size() * sizeof(float), c->stream));
    HIPCHK(c, c->mesh_p1.upload(p1.data(), p1.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->mesh_e1.upload(e1.data(), e1.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->mesh_e2.upload(e2.data(), e2.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->mesh_nrm.upload(nr.data(), nr.size() * sizeof(float4), c->stream));
    c->h_mesh_rec.resize((size_t)4 * n);
    for (int i = 0; i < n; i++) { c->h_mesh_rec[4 * (size_t)i] = p1[i]; c->h_mesh_rec[4 * (size_t)i + 1] = e1[i]; c->h_mesh_rec[4 * (size_t)i + 2] = e2[i]; c->h_mesh_rec[4 * (size_t)i + 3] = nr[i]; }
    c->prim_dirty = true;
    // BVH for the traversal variant: leaf-ordered (p1 | original index, e1, e2) records + 64-byte nodes
    Bvh bvh;
    bool finite = true; float scale = 0.0f;
    for (int i = 0; i < n && finite; i++)
        for (int k = 0; k < 3; k++) {
            const float v[3] = {tris[i].p1[k], tris[i].p2[k], tris[i].p3[k]};
            for (float x : v) { if (!std::isfinite(x)) finite = false; else scale = std::max(scale, std::fabs(x)); }
        }
    c->mesh_bvh_ok = false; c->nodes_dirty = true;
    std::vector<float4> btris;
    c->n_bvh_tris = c->n_list = c->n_never = 0;
    if (n > 0 && finite && scale < 1e18f) {
        // boxes grown by what the shader's float arithmetic makes of each triangle; the ones no box can stand for in leaves of
        // their own that every ray behind the gate reaches (pt_bvh.hpp classify_triangle, attach_listed)
        std::vector<float> grow(n);
        std::vector<unsigned char> cls(n);
        std::vector<int> listed;
        for (int i = 0; i < n; i++) {
            const float a[3] = {e1[i].x, e1[i].y, e1[i].z}, b2[3] = {e2[i].x, e2[i].y, e2[i].z};
            cls[i] = (unsigned char)classify_triangle(a, b2, scale, grow[i]);
            if (cls[i] == kTriListed) listed.push_back(i);
            else if (cls[i] == kTriNever) c->n_never++;
        }
        c->n_list = (int)listed.size();
        const int n_tree = n - c->n_list - c->n_never;
        // every listed triangle costs every ray behind the gate a test: the tree pays while they are few (kMeshListMax) or at
        // most half of the mesh -- beyond that the brute-force kernels are the better exact path (and the only one for a mesh
        // with nothing left for a tree)
        const bool usable = n_tree > 0 && (c->n_list <= kMeshListMax || c->n_list <= n / 2);
        if (usable) { build_bvh(tris, n, bvh, grow.data(), cls.data()); if ((int)bvh.order.size() == n_tree) attach_listed(bvh, listed); }
        const int n_rec = n_tree + c->n_list;
        if (usable && bvh.depth <= kBvhMaxDepth && (int)bvh.order.size() == n_rec) {
            c->n_bvh_tris = n_rec;
            btris.resize((size_t)kBvhTriVecs * n_rec);
            for (int j = 0; j < n_rec; j++) {
                const int i = bvh.order[j];
                float idf; std::memcpy(&idf, &i, 4);
                float4 *r = &btris[kBvhTriVecs * (size_t)j];
                r[0] = make_float4(p1[i].x, p1[i].y, p1[i].z, idf);
                r[1] = make_float4(e1[i].x, e1[i].y, e1[i].z, p1[i].w);      // | material (the canonical record keeps it in p1.w)
                r[2] = e2[i]; r[3] = nr[i];
            }
            HIPCHK(c, c->bvh_tris.upload(btris.data(), btris.size() * sizeof(float4), c->stream));
            c->h_mesh_nodes.swap(bvh.nodes);
            c->mesh_bvh_ok = true; c->mesh_scale = scale; c->mesh_depth = bvh.depth;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));    // host staging vectors die here
    c->n_mesh = n; c->n_chunks = chunks; c->max_mat_mesh = maxm;
    for (int k = 0; k < 3; k++) { c->bmin[k] = n ? nodes[0].boundsMin[k] : 0.0f; c->bmax[k] = n ? nodes[0].boundsMax[k] : 0.0f; }
    c->have_mesh = true; c->order_dirty = true;
    // tables the kernels read are brought up to date as soon as both halves of the scene are known, so that a render
    // (pt_render_static in particular) finds nothing left to upload
    if (c->have_scene) return sync_bvh(c);
    return PT_OK;
}

int pt_upload_scene(pt_ctx *c, const PtMaterial *mats, int n_mats, const PtSphere *sph, int n_sph, const PtTriangle *tris, int n_tris)
{
    if (!c) return PT_E_INVALID;
    if (n_mats < 0 || n_sph < 0 || n_tris < 0 || (n_mats && !mats) || (n_sph && !sph) || (n_tris && !tris))
        return fail(c, PT_E_INVALID, "bad scene arrays");
    if (n_sph > (1 << 20) || n_tris > (1 << 20) || n_mats > (1 << 20)) return fail(c, PT_E_LIMIT, "more than 2^20 scene records");
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<float4> p1(n_tris), e1(n_tris), e2(n_tris), nr(n_tris), sp(n_sph);
    std::vector<uint32_t> sm(n_sph);
    uint32_t maxm = 0;
    for (int i = 0; i < n_tris; i++) { tri_precompute(tris[i], p1[i], e1[i], e2[i], nr[i]); if (tris[i].materialIndex > maxm) maxm = tris[i].materialIndex; }
    c->h_tri_rec.resize((size_t)4 * n_tris);
    for (int i = 0; i < n_tris; i++) { c->h_tri_rec[4 * (size_t)i] = p1[i]; c->h_tri_rec[4 * (size_t)i + 1] = e1[i]; c->h_tri_rec[4 * (size_t)i + 2] = e2[i]; c->h_tri_rec[4 * (size_t)i + 3] = nr[i]; }
    c->prim_dirty = true;
    for (int i = 0; i < n_sph; i++) {
        sp[i] = make_float4(sph[i].position[0], sph[i].position[1], sph[i].position[2], sph[i].radius);
        sm[i] = sph[i].materialIndex; if (sm[i] > maxm) maxm = sm[i];
    }
    HIPCHK(c, c->tri_p1.upload(p1.data(), p1.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->tri_e1.upload(e1.data(), e1.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->tri_e2.upload(e2.data(), e2.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->tri_nrm.upload(nr.data(), nr.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->sph.upload(sp.data(), sp.size() * sizeof(float4), c->stream));
    HIPCHK(c, c->sph_mat.upload(sm.data(), sm.size() * sizeof(uint32_t), c->stream));
    HIPCHK(c, c->mats.upload(mats, (size_t)n_mats * sizeof(PtMaterial), c->stream));
    c->h_ior.resize((size_t)n_mats);
    for (int i = 0; i < n_mats; i++) c->h_ior[i] = mats[i].ior;
    { const int rc = upload_fresnel_table(c); if (rc != PT_OK) return rc; }
    // sphere tree for the traversal variants (records in leaf order: (centre, radius), (bits(index), 0, 0, 0))
    c->sph_bvh_ok = false; c->nodes_dirty = true;
    std::vector<float4> srec;
    {
        bool finite = true; float scale = 0.0f;
        for (int k = 0; k < 3; k++) { c->sph_bmin[k] = INFINITY; c->sph_bmax[k] = -INFINITY; }
        for (int i = 0; i < n_sph && finite; i++) {
            const float v[4] = {sph[i].position[0], sph[i].position[1], sph[i].position[2], sph[i].radius};
            for (float x : v) { if (!std::isfinite(x)) finite = false; }
            for (int k = 0; k < 3; k++) {
                c->sph_bmin[k] = std::min(c->sph_bmin[k], v[k] - std::fabs(v[3])); c->sph_bmax[k] = std::max(c->sph_bmax[k], v[k] + std::fabs(v[3]));
            }
            for (int k = 0; k < 3; k++) scale = std::max(scale, std::fabs(sph[i].position[k]) + std::fabs(sph[i].radius));
        }
        c->sph_rmin = INFINITY; c->sph_reach = 0.0f;
        for (int i = 0; i < n_sph && finite; i++) {
            c->sph_rmin = std::min(c->sph_rmin, std::fabs(sph[i].radius));
            c->sph_reach = std::max(c->sph_reach, (std::fabs(sph[i].position[0]) + std::fabs(sph[i].position[1])) + std::fabs(sph[i].position[2]));
        }
        if (!(c->sph_rmin < INFINITY)) c->sph_rmin = 0.0f;
        c->sph_rmin = std::max(c->sph_rmin, 1e-30f);      // the margin divides by sqrt(r^2 + k D^2) + r: never 0 / 0 (a point sphere seen from its centre)
        if (n_sph > 0 && finite && scale < 1e18f) {
            Bvh bvh;
            build_sphere_bvh(sph, n_sph, bvh);
            if (bvh.depth <= kBvhMaxDepth && (int)bvh.order.size() == n_sph) {
                srec.resize((size_t)2 * n_sph);
                for (int j = 0; j < n_sph; j++) {
                    const int i = bvh.order[j];
                    float idf; std::memcpy(&idf, &i, 4);
                    srec[2 * (size_t)j] = sp[i];
                    srec[2 * (size_t)j + 1] = make_float4(idf, 0.0f, 0.0f, 0.0f);
                }
                HIPCHK(c, c->bvh_sph.upload(srec.data(), srec.size() * sizeof(float4), c->stream));
                c->h_sph_nodes.swap(bvh.nodes);
                c->sph_bvh_ok = true; c->sph_scale = scale; c->sph_depth = bvh.depth;
            }
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_tris = n_tris; c->n_sph = n_sph; c->n_mats = n_mats; c->max_mat_scene = maxm;
    c->have_scene = true; c->order_dirty = true;
    if (c->have_mesh) return sync_bvh(c);
    return PT_OK;
}

int pt_set_camera(pt_ctx *c, const float pos[3], const float iv[16])
{
    if (!c || !pos || !iv) return PT_E_INVALID;
    for (int k = 0; k < 3; k++) c->cam[k] = pos[k];
    for (int col = 0; col < 3; col++)
        for (int r = 0; r < 3; r++) c->iv[col * 3 + r] = iv[col * 4 + r];
    c->have_camera = true; c->order_dirty = true;
    return PT_OK;
}

int pt_set_debug_normal(pt_ctx *c, int enabled) { if (!c) return PT_E_INVALID; c->debug_normal = enabled ? 1 : 0; return PT_OK; }

int pt_set_max_bounces(pt_ctx *c, int n)
{
    if (!c) return PT_E_INVALID;
    if (n < 1 || n > 4096) return fail(c, PT_E_INVALID, "max_bounces %d outside [1,4096]", n);
    c->max_bounces = n;
    return PT_OK;
}

int pt_shading_preset(PtShadingParams *p, int preset)
{
    if (!p || preset < 0 || preset > 1) return PT_E_INVALID;
    std::memset(p, 0, sizeof *p);
    if (preset == 0) {            // res/shaders/rt.comp:10,25-26,305,332
        p->hit_limit = kHitLimit; p->origin_offset = 0.0002f; p->jitter_scale = 0.00025f;
        p->sky_color[0] = p->sky_color[1] = p->sky_color[2] = 1.0f; p->sky_intensity = 0.5f;
        p->trace_mesh = 1;
    } else {                      // res/shaders/rt.frag:131,166,253-254,283,290-295,303,320,332
        p->hit_limit = 0.0f; p->origin_offset = 0.00015f; p->jitter_scale = 0.0005f;
        p->sky_color[0] = 0.92f; p->sky_color[1] = 0.8f; p->sky_color[2] = 0.85f; p->sky_intensity = 0.8f;
        p->fresnel_n1_one = 1; p->two_hemisphere_draws = 1; p->pixel_center = 1; p->sqrt_output = 1; p->trace_mesh = 0; p->sphere_sum_xyz = 1;
    }
    return PT_OK;
}

int pt_set_shading_params(pt_ctx *c, const PtShadingParams *p)
{
    if (!c) return PT_E_INVALID;
    if (!p) return fail(c, PT_E_INVALID, "params is NULL");
    const float v[7] = {p->hit_limit, p->origin_offset, p->jitter_scale, p->sky_color[0], p->sky_color[1], p->sky_color[2], p->sky_intensity};
    for (float x : v) if (!std::isfinite(x)) return fail(c, PT_E_INVALID, "shading parameters must be finite");
    if (p->hit_limit < 0.0f || p->hit_limit > 1e-3f) return fail(c, PT_E_INVALID, "hit_limit %g outside [0, 1e-3]", (double)p->hit_limit);
    if (p->origin_offset < 0.0f || p->jitter_scale < 0.0f) return fail(c, PT_E_INVALID, "origin_offset and jitter_scale must not be negative");
    const int32_t flags[6] = {p->fresnel_n1_one, p->two_hemisphere_draws, p->pixel_center, p->sqrt_output, p->trace_mesh, p->sphere_sum_xyz};
    for (int32_t f : flags) if (f != 0 && f != 1) return fail(c, PT_E_INVALID, "shading flags are 0 or 1");
    const bool table = p->fresnel_n1_one != c->shading.fresnel_n1_one;
    c->shading = *p;
    c->order_dirty = true;        // the tile schedule looks at pixel centres and at whether the mesh is traced
    if (table && c->have_scene) {
        Bind b(c); HIPCHK(c, b.e);
        HIPCHK(c, hipStreamSynchronize(c->stream));                 // no launch may still be reading the old table
        return upload_fresnel_table(c);
    }
    return PT_OK;
}

int pt_get_shading_params(pt_ctx *c, PtShadingParams *out)
{
    if (!c || !out) return PT_E_INVALID;
    *out = c->shading;
    return PT_OK;
}

int pt_render_pass(pt_ctx *c, int pass)
{
    if (!c) return PT_E_INVALID;
    if (pass < 1) return fail(c, PT_E_INVALID, "currAccumPass starts at 1 (src/Main.cpp:144); pass 0 would divide by zero");
    return render_checked(c, pass, 1, 0);
}

int pt_render(pt_ctx *c, int first_pass, int n_passes)
{
    if (!c) return PT_E_INVALID;
    if (first_pass < 1) return fail(c, PT_E_INVALID, "currAccumPass starts at 1 (src/Main.cpp:144)");
    return render_checked(c, first_pass, n_passes, 0);
}

int pt_render_static(pt_ctx *c, int first_pass, int n_passes)
{
    if (!c) return PT_E_INVALID;
    if (first_pass < 1) return fail(c, PT_E_INVALID, "currAccumPass starts at 1 (src/Main.cpp:144)");
    return launch_trace(c, first_pass, n_passes, 0, true);
}

int pt_render_sum(pt_ctx *c, int first_pass, int n_passes) { if (!c) return PT_E_INVALID; return render_checked(c, first_pass, n_passes, 1); }

int pt_scale_accum(pt_ctx *c, float scale, int set_alpha_one)
{
    if (!c) return PT_E_INVALID;
    if (!c->accum) return fail(c, PT_E_STATE, "image size not set (pt_resize)");
    Bind b(c); HIPCHK(c, b.e);
    size_t n = (size_t)c->width * c->height;
    hipLaunchKernelGGL(pt_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (float4 *)c->accum, n, scale, set_alpha_one);
    HIPCHK(c, hipGetLastError());
    return PT_OK;
}

int pt_reset(pt_ctx *c)
{
    if (!c) return PT_E_INVALID;
    if (!c->accum) return fail(c, PT_E_STATE, "image size not set (pt_resize)");
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipMemsetAsync(c->accum, 0, (size_t)c->width * c->height * 4 * sizeof(float), c->stream));
    return PT_OK;
}

int pt_synchronize(pt_ctx *c)
{
    if (!c) return PT_E_INVALID;
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_read_accum(pt_ctx *c, float *rgba, size_t bytes)
{
    if (!c || !rgba) return PT_E_INVALID;
    if (!c->accum) return fail(c, PT_E_STATE, "image size not set (pt_resize)");
    size_t need = (size_t)c->width * c->height * 4 * sizeof(float);
    if (bytes < need) return fail(c, PT_E_INVALID, "host buffer too small: %zu < %zu", bytes, need);
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipMemcpyAsync(rgba, c->accum, need, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

void *pt_accum_device_ptr(pt_ctx *c) { return c ? (void *)c->accum : nullptr; }

int pt_read_primary_hits(pt_ctx *c, int pass, int32_t *ids, float *t, float *dirs, uint32_t *rng)
{
    int rc = check_ready(c);
    if (rc) return rc;
    Bind b(c); HIPCHK(c, b.e);
    if ((rc = sync_bvh(c)) != PT_OK) return rc;
    const size_t n = (size_t)c->width * c->height;
    int32_t *d_ids = nullptr; float *d_t = nullptr, *d_dirs = nullptr; uint32_t *d_rng = nullptr;
    auto cleanup = [&]() { if (d_ids) (void)hipFree(d_ids); if (d_t) (void)hipFree(d_t); if (d_dirs) (void)hipFree(d_dirs); if (d_rng) (void)hipFree(d_rng); };
#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail(c, PT_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } } while (0)
    if (ids) { TRY(hipMalloc((void **)&d_ids, n * 4)); TRY(hipMemsetAsync(d_ids, 0xff, n * 4, c->stream)); }
    if (t) { TRY(hipMalloc((void **)&d_t, n * 4)); TRY(hipMemsetAsync(d_t, 0, n * 4, c->stream)); }
    if (dirs) { TRY(hipMalloc((void **)&d_dirs, n * 12)); TRY(hipMemsetAsync(d_dirs, 0, n * 12, c->stream)); }
    if (rng) { TRY(hipMalloc((void **)&d_rng, n * 4)); TRY(hipMemsetAsync(d_rng, 0, n * 4, c->stream)); }
    SceneView sv = scene_view(c);
    FrameParams fp = frame_params(c, pass, 1, 0);
    fp.counters = nullptr;
    fp.out_ids = d_ids; fp.out_t = d_t; fp.out_dirs = d_dirs; fp.out_rng = d_rng;
    if (pick_variant(c) >= 5)
        hipLaunchKernelGGL(pt_primary_hits_bvh<kSlots>, dim3((unsigned)tile_count(c->width, c->height)), dim3(kWave), 0, c->stream, sv, fp);
    else
        hipLaunchKernelGGL(pt_primary_hits<kSlots>, dim3((unsigned)tile_count(c->width, c->height)), dim3(kWave), 0, c->stream, sv, fp);
    TRY(hipGetLastError());
    if (ids) TRY(hipMemcpyAsync(ids, d_ids, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (t) TRY(hipMemcpyAsync(t, d_t, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (dirs) TRY(hipMemcpyAsync(dirs, d_dirs, n * 12, hipMemcpyDeviceToHost, c->stream));
    if (rng) TRY(hipMemcpyAsync(rng, d_rng, n * 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
#undef TRY
    cleanup();
    return PT_OK;
}

int pt_get_counters(pt_ctx *c, PtCounters *out)
{
    if (!c || !out) return PT_E_INVALID;
    Bind b(c); HIPCHK(c, b.e);
    unsigned long long v[kNumCounters];
    HIPCHK(c, hipMemcpyAsync(v, c->counters.p, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->samples = v[0]; out->segments = v[1]; out->mesh_segments = v[2]; out->mesh_tri_tests = v[3]; out->hits = v[4]; out->wave_iterations = v[5];
    out->primary_mesh_segments = v[6]; out->waves = v[7]; out->wave_ticks_sum = v[8]; out->wave_ticks_max = v[9];
    out->bvh_node_visits = v[10]; out->bvh_tri_tests = v[11];
    out->phase_cycles[0] = v[12]; out->phase_cycles[1] = v[13]; out->phase_cycles[2] = v[14]; out->phase_cycles[3] = v[15];
    out->bvh_sphere_tests = v[16]; out->bvh_box_tests = v[17];
    return PT_OK;
}

int pt_reset_counters(pt_ctx *c)
{
    if (!c) return PT_E_INVALID;
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipMemsetAsync(c->counters.p, 0, kNumCounters * sizeof(unsigned long long), c->stream));
    return PT_OK;
}

int pt_last_render_ms(pt_ctx *c, float *ms, int *n_launches)
{
    if (!c || !ms) return PT_E_INVALID;
    if (c->n_timed < 1) return fail(c, PT_E_STATE, "no timed render on this stream yet (pt_render_static records no timing events)");
    Bind b(c); HIPCHK(c, b.e);
    float total = 0.0f;
    for (int i = 0; i < c->n_timed; i++) {
        float t = 0.0f;
        HIPCHK(c, hipEventSynchronize(c->ev[2 * i + 1]));
        HIPCHK(c, hipEventElapsedTime(&t, c->ev[2 * i], c->ev[2 * i + 1]));
        total += t;
    }
    *ms = total;
    if (n_launches) *n_launches = c->n_timed;
    return PT_OK;
}

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------
extern "C++" {
namespace {
struct Rccl {
    typedef struct { char internal[PT_GROUP_ID_BYTES]; } UniqueId;            // ncclUniqueId (rccl.h:43)
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { r.why = std::string("librccl.so not loadable: ") + dlerror(); return r; }
    r.GetUniqueId = (int (*)(Rccl::UniqueId *))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void **, int, Rccl::UniqueId, int))dlsym(h, "ncclCommInitRank");
    r.Reduce = (int (*)(const void *, void *, size_t, int, int, int, void *, hipStream_t))dlsym(h, "ncclReduce");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    r.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.Reduce && r.AllReduce && r.CommDestroy && r.GetErrorString;
    if (!r.ok) r.why = "librccl.so lacks an expected symbol";
    return r;
}
constexpr int kNcclFloat32 = 7, kNcclInt32 = 2, kNcclSum = 0, kNcclMax = 2;     // rccl.h ncclDataType_t / ncclRedOp_t
constexpr int kGroupAgree = 15;            // seven shared arguments as (x, -x) pairs + one status word
// 31-bit FNV-1a of the shading parameters (pt_group_render: every rank must shade alike)
int shading_hash(const PtShadingParams &p)
{
    const float fl[7] = {p.hit_limit, p.origin_offset, p.jitter_scale, p.sky_color[0], p.sky_color[1], p.sky_color[2], p.sky_intensity};
    const int32_t in[6] = {p.fresnel_n1_one, p.two_hemisphere_draws, p.pixel_center, p.sqrt_output, p.trace_mesh, p.sphere_sum_xyz};
    uint32_t h = 2166136261u;
    auto eat = [&](const void *q, size_t n) { for (size_t i = 0; i < n; i++) h = (h ^ ((const unsigned char *)q)[i]) * 16777619u; };
    eat(fl, sizeof fl); eat(in, sizeof in);
    return (int)(h & 0x7fffffffu);
}
}  // namespace
}  // extern "C++"

int pt_group_unique_id(char id[PT_GROUP_ID_BYTES])
{
    if (!id) return PT_E_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return fail(nullptr, PT_E_STATE, "%s", r.why.c_str());
    Rccl::UniqueId u;
    const int rc = r.GetUniqueId(&u);
    if (rc) return fail(nullptr, PT_E_HIP, "ncclGetUniqueId: %s", r.GetErrorString(rc));
    std::memcpy(id, u.internal, PT_GROUP_ID_BYTES);
    return PT_OK;
}

int pt_group_init(pt_ctx *c, const char id[PT_GROUP_ID_BYTES], int rank, int world)
{
    if (!c || !id) return PT_E_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, PT_E_INVALID, "bad rank %d / world %d", rank, world);
    if (c->group_comm) return fail(c, PT_E_STATE, "the context already belongs to a group (pt_group_finalize first)");
    Rccl &r = rccl();
    if (!r.ok) return fail(c, PT_E_STATE, "%s", r.why.c_str());
    Bind b(c); HIPCHK(c, b.e);
    Rccl::UniqueId u;
    std::memcpy(u.internal, id, PT_GROUP_ID_BYTES);
    const int rc = r.CommInitRank(&c->group_comm, world, u, rank);
    if (rc) { c->group_comm = nullptr; return fail(c, PT_E_HIP, "ncclCommInitRank: %s", r.GetErrorString(rc)); }
    c->group_rank = rank; c->group_world = world;
    if (!c->group_agree) HIPCHK(c, hipMalloc((void **)&c->group_agree, kGroupAgree * sizeof(int)));
    return PT_OK;
}

int pt_group_render(pt_ctx *c, int first_pass, int total_spp)
{
    if (!c) return PT_E_INVALID;
    if (!c->group_comm) return fail(c, PT_E_STATE, "pt_group_init first");
    Bind b(c); HIPCHK(c, b.e);
    Rccl &r = rccl();
    // A collective must not be left by one rank alone: whatever is wrong locally (bad arguments, scene not uploaded) is
    // first shared with the other ranks -- ONE all-reduce (max) of fifteen ints: the arguments every rank must agree on, each
    // together with its negative (max of -x = -min of x), and this rank's local status -- and only then does anybody return.
    int local = PT_OK;
    if (first_pass < 1 || total_spp < 1) local = fail(c, PT_E_INVALID, "bad pass range %d+%d", first_pass, total_spp);
    if (local == PT_OK) local = check_ready(c);
    constexpr int kShared = (kGroupAgree - 1) / 2;
    const int mine[kShared] = {first_pass, total_spp, c->width, c->height, c->max_bounces, c->debug_normal, shading_hash(c->shading)};
    int v[kGroupAgree] = {0};
    for (int k = 0; k < kShared; k++) { v[2 * k] = mine[k]; v[2 * k + 1] = -mine[k]; }
    v[kGroupAgree - 1] = -local;                                  // PT_E_* are negative: max over ranks of -status > 0 means some rank failed
    HIPCHK(c, hipMemcpyAsync(c->group_agree, v, sizeof v, hipMemcpyHostToDevice, c->stream));
    int nrc = r.AllReduce(c->group_agree, c->group_agree, kGroupAgree, kNcclInt32, kNcclMax, c->group_comm, c->stream);
    if (nrc) return fail(c, PT_E_HIP, "ncclAllReduce: %s", r.GetErrorString(nrc));
    int all[kGroupAgree];
    HIPCHK(c, hipMemcpyAsync(all, c->group_agree, sizeof all, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (local != PT_OK) return local;                // this rank's own message is already set (the others leave two lines below)
    if (all[kGroupAgree - 1] > 0) return fail(c, PT_E_STATE, "another rank of the group failed its checks (status %d); nothing was rendered", -all[kGroupAgree - 1]);
    // every rank sees the same (max, min) pairs, so either all of them return here or none does
    static const char *what[kShared] = {"first_pass", "total_spp", "width", "height", "max_bounces", "debug_normal", "the shading parameters (hash)"};
    for (int k = 0; k < kShared; k++)
        if (all[2 * k] != -all[2 * k + 1])
            return fail(c, PT_E_INVALID, "the ranks of the group disagree on %s (here %d, over the group %d..%d); nothing was rendered", what[k], mine[k], -all[2 * k + 1], all[2 * k]);
    // contiguous, balanced split (the first total % world ranks take one pass more): path_tracer_amd/distributed.py pass_range
    const int base = total_spp / c->group_world, extra = total_spp % c->group_world;
    const int count = base + (c->group_rank < extra ? 1 : 0);
    const int first = first_pass + c->group_rank * base + std::min(c->group_rank, extra);
    const size_t n_floats = (size_t)c->width * c->height * 4;
    HIPCHK(c, hipMemsetAsync(c->accum, 0, n_floats * sizeof(float), c->stream));
    int rc = PT_OK;
    if (count > 0) rc = launch_trace(c, first, count, 1);
    if (rc != PT_OK) {
        // the peers are already on their way into the reduce: take part with an all-zero image, report the failure afterwards
        (void)hipMemsetAsync(c->accum, 0, n_floats * sizeof(float), c->stream);
    }
    nrc = r.Reduce(c->accum, c->accum, n_floats, kNcclFloat32, kNcclSum, 0, c->group_comm, c->stream);
    // Second status round, behind the reduce on the same stream: a rank whose trace failed AFTER the agreement round joined
    // the reduce with zeros, and without this only that rank would know -- rank 0 would scale a sum that lacks its samples
    // and return PT_OK with a darker image.  One int, all-reduce (max) of -status; every rank returns an error if any failed.
    const int mine2 = -(rc != PT_OK ? rc : (nrc ? PT_E_HIP : PT_OK));
    int nrc2 = 0, worst = 0;
    hipError_t he = hipMemcpyAsync(c->group_agree, &mine2, sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) nrc2 = r.AllReduce(c->group_agree, c->group_agree, 1, kNcclInt32, kNcclMax, c->group_comm, c->stream);
    if (he == hipSuccess && !nrc2) he = hipMemcpyAsync(&worst, c->group_agree, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    // rank 0 also looks at the sample count the sum carries in alpha (every pass adds 1 there, pt_render_sum)
    float alpha0 = (float)total_spp;
    if (he == hipSuccess && c->group_rank == 0) he = hipMemcpyAsync(&alpha0, c->accum + 3, sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (rc != PT_OK) return rc;                      // this rank's own failure keeps its own message
    if (nrc) return fail(c, PT_E_HIP, "ncclReduce: %s", r.GetErrorString(nrc));
    if (nrc2) return fail(c, PT_E_HIP, "ncclAllReduce (status): %s", r.GetErrorString(nrc2));
    if (he != hipSuccess) return fail(c, PT_E_HIP, "status round after the reduce: %s", hipGetErrorString(he));
    if (worst > 0) return fail(c, PT_E_STATE, "another rank of the group failed while tracing (status %d): the image on rank 0 lacks its samples", -worst);
    if (c->group_rank == 0 && alpha0 != (float)total_spp)
        return fail(c, PT_E_STATE, "the reduced image carries %.9g samples per pixel, %d were asked for", (double)alpha0, total_spp);
    if (c->group_rank == 0) return pt_scale_accum(c, 1.0f / (float)total_spp, 1);
    return PT_OK;
}

int pt_group_finalize(pt_ctx *c)
{
    if (!c) return PT_E_INVALID;
    if (!c->group_comm) return PT_OK;
    Bind b(c); HIPCHK(c, b.e);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int rc = rccl().CommDestroy(c->group_comm);
    c->group_comm = nullptr; c->group_world = 0; c->group_rank = 0;
    if (rc) return fail(c, PT_E_HIP, "ncclCommDestroy: %s", rccl().GetErrorString(rc));
    return PT_OK;
}

int pt_classify_mesh(const PtTriangle *tris, int n, int *in_tree, int *listed, int *never, float *max_growth)
{
    if (n < 0 || (n > 0 && !tris)) return PT_E_INVALID;
    float scale = 0.0f;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            const float v[3] = {tris[i].p1[k], tris[i].p2[k], tris[i].p3[k]};
            for (float x : v) { if (!std::isfinite(x)) return PT_E_INVALID; scale = std::max(scale, std::fabs(x)); }
        }
    int cnt[3] = {0, 0, 0};
    float big = 0.0f;
    for (int i = 0; i < n; i++) {
        float4 p1, e1, e2, nr;
        tri_precompute(tris[i], p1, e1, e2, nr);
        const float a[3] = {e1.x, e1.y, e1.z}, b[3] = {e2.x, e2.y, e2.z};
        float g;
        const TriClass c = classify_triangle(a, b, scale, g);
        cnt[c]++;
        if (c == kTriTree) big = std::max(big, g);
    }
    if (in_tree) *in_tree = cnt[kTriTree];
    if (listed) *listed = cnt[kTriListed];
    if (never) *never = cnt[kTriNever];
    if (max_growth) *max_growth = big;
    return PT_OK;
}

int pt_bvh_selftest(const PtTriangle *tris, int n, int *n_nodes, int *depth, int *max_leaf, float *sah_cost)
{
    if (!tris || n < 1) return PT_E_INVALID;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            if (!std::isfinite(tris[i].p1[k]) || !std::isfinite(tris[i].p2[k]) || !std::isfinite(tris[i].p3[k])) return PT_E_INVALID;
    Bvh bvh;
    build_bvh(tris, n, bvh);
    if (bvh.nodes.empty() || (int)bvh.order.size() != n || bvh.depth > kBvhMaxDepth) return PT_E_STATE;
    std::vector<int> seen(n, 0);
    int biggest = 0, deepest = 0;
    double cost = 0.0;
    bool ok = true;
    // returns the box of the subtree under `code`, checking containment on the way up
    struct Walker {
        const Bvh &b; const PtTriangle *tris; int n; std::vector<int> &seen; int &biggest, &deepest; double &cost; bool &ok;
        bvh_detail::Box walk(int code, int d)
        {
            bvh_detail::Box box; box.reset();
            if (d > kBvhMaxDepth + 1) { ok = false; return box; }
            deepest = std::max(deepest, d);
            if (code < 0) {
                const int c = ~code, first = c >> 3, cnt = (c & 7) + 1;
                if ((c & kSphereLeaf) || first < 0 || first + cnt > n) { ok = false; return box; }
                biggest = std::max(biggest, cnt);
                for (int j = first; j < first + cnt; j++) {
                    const int i = b.order[j];
                    if (i < 0 || i >= n) { ok = false; return box; }
                    seen[i]++;
                    box.grow(tris[i].p1); box.grow(tris[i].p2); box.grow(tris[i].p3);
                }
                cost += box.half_area() * cnt;
                return box;
            }
            if (code >= (int)b.nodes.size()) { ok = false; return box; }
            const BvhNode &nd = b.nodes[code];
            const bvh_detail::Box l = walk(nd.left, d + 1), r = walk(nd.right, d + 1);
            for (int k = 0; k < 3; k++) {
                if (!(nd.lmin[k] <= l.mn[k] && nd.lmax[k] >= l.mx[k] && nd.rmin[k] <= r.mn[k] && nd.rmax[k] >= r.mx[k])) ok = false;
            }
            box.grow(l); box.grow(r);
            cost += box.half_area();
            return box;
        }
    } w{bvh, tris, n, seen, biggest, deepest, cost, ok};
    const bvh_detail::Box root = w.walk(0, 0);
    // a mesh of <= kLeafMax triangles has a stand-in root whose two children are the same leaf: each triangle is seen twice
    const int expect = n <= kLeafMax ? 2 : 1;
    for (int i = 0; i < n; i++) if (seen[i] != expect) ok = false;
    if (!ok) return PT_E_STATE;
    // the four-wide collapse the kernels traverse (collapse_bvh4): the same leaves, each reached exactly once; every child box
    // contains its subtree; unused slots carry the far box and kBvhNoChild; the stack bound covers the deepest chain of pushes
    {
        Bvh4 w4;
        collapse_bvh4(bvh.nodes, w4, 0);
        if (w4.nodes.empty() || w4.nodes.size() > bvh.nodes.size()) return PT_E_STATE;
        std::vector<int> seen4(n, 0);
        bool ok4 = true;
        struct Walk4 {
            const Bvh4 &w; const Bvh &b; const PtTriangle *tris; int n; std::vector<int> &seen; bool &ok; int max_push = 0;
            bvh_detail::Box walk(int code, int pushed, int d)
            {
                bvh_detail::Box box; box.reset();
                if (d > kBvhMaxDepth + 1) { ok = false; return box; }
                if (code < 0) {
                    const int c = ~code, first = c >> 3, cnt = (c & 7) + 1;
                    if (code == kBvhNoChild || first < 0 || first + cnt > n) { ok = false; return box; }
                    for (int j = first; j < first + cnt; j++) { const int i = b.order[j]; seen[i]++; box.grow(tris[i].p1); box.grow(tris[i].p2); box.grow(tris[i].p3); }
                    return box;
                }
                if (code >= (int)w.nodes.size()) { ok = false; return box; }
                const BvhNode4 &nd = w.nodes[code];
                int used = 0;
                for (int i = 0; i < 4; i++) if (nd.child[i] != kBvhNoChild) used++;
                if (used < 2) ok = false;
                max_push = std::max(max_push, pushed + used - 1);
                for (int i = 0; i < 4; i++) {
                    if (nd.child[i] == kBvhNoChild) { if (nd.mnx[i] != kBvhFarBox || nd.mxz[i] != kBvhFarBox || i < used) ok = false; continue; }
                    const bvh_detail::Box sub = walk(nd.child[i], pushed + used - 1, d + 1);
                    const float mn[3] = {nd.mnx[i], nd.mny[i], nd.mnz[i]}, mx[3] = {nd.mxx[i], nd.mxy[i], nd.mxz[i]};
                    for (int k = 0; k < 3; k++) if (!(mn[k] <= sub.mn[k] && mx[k] >= sub.mx[k])) ok = false;
                    box.grow(sub);
                }
                return box;
            }
        } w4k{w4, bvh, tris, n, seen4, ok4};
        w4k.walk(0, 0, 0);
        for (int i = 0; i < n; i++) if (seen4[i] != expect) ok4 = false;
        if (!ok4 || w4k.max_push != w4.stack_need) return PT_E_STATE;
        // the 64-byte quantised form (read in place by the global-memory kernel): every plane a lane rebuilds,
        // fmaf(q, 2^e, o), must lie on or outside the float plane it stands for, and the codes must be the float node's
        for (const BvhNode4 &nd : w4.nodes) {
            BvhNode4Q q;
            if (!quantise_bvh4(nd, q)) return PT_E_STATE;
            const float *mn[3] = {nd.mnx, nd.mny, nd.mnz}, *mx[3] = {nd.mxx, nd.mxy, nd.mxz};
            for (int k = 0; k < 3; k++) {
                const uint32_t eb = (q.exps >> (8 * k)) & 0xffu;
                float step; const uint32_t sb = eb << 23; std::memcpy(&step, &sb, 4);
                for (int i = 0; i < 4; i++) {
                    if (q.child[i] != nd.child[i]) return PT_E_STATE;
                    if (nd.child[i] == kBvhNoChild) continue;
                    const float lo = std::fmaf((float)q.qmn[k][i], step, q.o[k]), hi = std::fmaf((float)q.qmx[k][i], step, q.o[k]);
                    if (!(lo <= mn[k][i] && hi >= mx[k][i])) return PT_E_STATE;
                    // ... and not by more than two steps (the quantisation must stay tight enough to cull)
                    if (!(mn[k][i] - lo <= 2.0f * step && hi - mx[k][i] <= 2.0f * step)) return PT_E_STATE;
                }
            }
        }
    }
    if (n_nodes) *n_nodes = (int)bvh.nodes.size();
    if (depth) *depth = bvh.depth;
    if (max_leaf) *max_leaf = biggest;
    if (sah_cost) *sah_cost = root.half_area() > 0.0f ? (float)(cost / root.half_area()) : 0.0f;
    return PT_OK;
}

int pt_set_option(pt_ctx *c, const char *name, int value)
{
    if (!c || !name) return PT_E_INVALID;
    struct { const char *name; int *slot; int lo, hi; } opts[] = {
        {"park_high", &c->opt_park_high, 0, 64}, {"park_wait", &c->opt_park_wait, 0, 1 << 20}, {"refill_min", &c->opt_refill_min, 0, 64},
        {"park_exit", &c->opt_park_exit, 0, 8}, {"ppi1_mul", &c->opt_ppi1_mul, 1, 4096}, {"ppi2_mul", &c->opt_ppi2_mul, 1, 4096},
        {"no_fused", &c->opt_no_fused, 0, 1}, {"wgs_per_cu", &c->opt_wgs_per_cu, 0, 8}, {"lds_block", &c->opt_lds_block, 0, 1024},
        {"verify", &c->opt_verify, 0, 1 << 30}, {"verify_inject", &c->opt_verify_inject, 0, 1},
    };
    for (auto &o : opts)
        if (!std::strcmp(name, o.name)) {
            if (value < o.lo || value > o.hi) return fail(c, PT_E_INVALID, "option %s = %d outside [%d,%d]", name, value, o.lo, o.hi);
            if (o.slot == &c->opt_lds_block && value != 0 && value != kBvhLdsBlock && value != kBvhLdsBlockWide)
                return fail(c, PT_E_INVALID, "option lds_block = %d: 0 (automatic), %d or %d", value, kBvhLdsBlock, kBvhLdsBlockWide);
            if (o.slot == &c->opt_verify_inject && value != *o.slot) c->nodes_dirty = true;     // the tree is uploaded again, with or without the fault
            *o.slot = value;
            return PT_OK;
        }
    return fail(c, PT_E_INVALID, "unknown option '%s'", name);
}

int pt_set_tuning(pt_ctx *c, int passes_per_item, size_t workspace_limit_bytes, int waves_per_cu, int variant)
{
    if (!c) return PT_E_INVALID;
    if (variant < 0 || variant > 6) return fail(c, PT_E_INVALID, "variant outside [0,6]");
    if (variant) c->variant = variant;
    if (waves_per_cu < 0 || waves_per_cu > 32) return fail(c, PT_E_INVALID, "waves_per_cu outside [1,32]");
    if (waves_per_cu) c->waves_per_cu = waves_per_cu;
    if (passes_per_item < 0 || passes_per_item > 4096) return fail(c, PT_E_INVALID, "passes_per_item outside [1,4096]");
    if (passes_per_item) c->chunk_passes = passes_per_item;
    if (workspace_limit_bytes) c->workspace_limit = workspace_limit_bytes;
    return PT_OK;
}

}  // extern "C"
