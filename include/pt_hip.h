/*
 * pt_hip.h — C ABI of the MI355X-native path-tracing hot path (libpt_hip.so).
 *
 * This is the drop-in boundary for the reference's per-pixel path-tracing loop
 * (res/shaders/rt.comp, dispatched at src/Main.cpp:148).  The reference has no plugin API; its
 * de-facto operator interface is the set of GL calls the host makes to feed and launch that
 * shader.  Every entry point below names the reference call site it replaces.
 *
 * Conventions: plain C, opaque handle, every call returns 0 on success or a negative PT_E_* code;
 * pt_last_error() gives the text.  Records are the reference's own host structs, byte for byte
 * (src/Object.h:8-75 == std430 layout in rt.comp:33-48); uploads copy, the caller keeps ownership.
 * One host thread per context; calls on one context are not re-entrant (the reference drives its
 * GL context from one thread, src/Main.cpp:108-170).  Work is enqueued on the context's HIP
 * stream; pt_read_* and pt_synchronize wait for it.
 * There is NO CPU fallback: every entry point fails if no HIP device is usable.
 */
#ifndef PT_HIP_H
#define PT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_E_INVALID (-1)   /* bad argument / call order */
#define PT_E_HIP (-2)       /* a HIP runtime call failed */
#define PT_E_STATE (-3)     /* scene, camera or image size not set yet */
#define PT_E_LIMIT (-4)     /* size beyond a documented maximum */

/* src/Object.h:8-21  (80 B; rt.comp:33-44) */
typedef struct PtMaterial {
    float baseColor[4], specularColor[4], emissionColor[4];
    float smoothness, specularSmoothness, emissionStrength, ior, refractionAmount, specularChance;
    int32_t pad[2];
} PtMaterial;
/* src/Object.h:23-33  (32 B; rt.comp:45) */
typedef struct PtSphere { float position[3]; float radius; uint32_t materialIndex; int32_t pad[3]; } PtSphere;
/* src/Object.h:35-48  (64 B; rt.comp:46) */
typedef struct PtTriangle { float p1[4], p2[4], p3[4]; uint32_t materialIndex; int32_t pad[3]; } PtTriangle;
/* src/Object.h:63-75  (48 B; rt.comp:48) */
typedef struct PtNode { float boundsMin[4], boundsMax[4]; int32_t triIndex, numTris, childrenIndex, pad; } PtNode;

typedef struct pt_ctx pt_ctx;

/* Work actually executed by the kernels since the last pt_reset_counters (for the roofline). */
typedef struct PtCounters {
    uint64_t samples;        /* primary rays traced                                             */
    uint64_t segments;       /* CalculateRay() calls (rt.comp:271)                              */
    uint64_t mesh_segments;  /* segments that passed the root-AABB gate (rt.comp:213)           */
    uint64_t mesh_tri_tests; /* ray-triangle tests against mesh triangles, padded slots included */
    uint64_t hits;           /* segments that hit something and were shaded (rt.comp:273-311)   */
    uint64_t wave_iterations;/* bounce-loop iterations executed by waves (64 lanes each)        */
    uint64_t primary_mesh_segments; /* mesh_segments that were primary rays                      */
    uint64_t waves;          /* persistent waves that ran                                       */
    uint64_t wave_ticks_sum; /* sum over waves of (last item done - wave start), 100 MHz ticks  */
    uint64_t wave_ticks_max; /* the longest such span: sum / (waves * max) = load balance       */
    uint64_t bvh_node_visits;/* BVH variant: inner nodes fetched, per lane (see bvh_box_tests)      */
    uint64_t bvh_tri_tests;  /* BVH variant: ray-triangle tests executed, per lane               */
    uint64_t phase_cycles[4];/* diagnostic builds (-DPT_PHASE_TIMERS=1): wave cycles in ray generation / closest hit / shading / (part of closest hit) mesh traversal; else 0 */
    uint64_t bvh_sphere_tests;/* BVH variant: ray-sphere tests executed in sphere-tree leaves (not part of bvh_tri_tests) */
    uint64_t bvh_box_tests;  /* BVH variant: ray-box (slab) tests executed: 4 per visit of a four-wide node (2 in a PT_BVH_WIDE=0 build) */
} PtCounters;

/* --- lifetime: Renderer::Init (src/Renderer.cpp:33-140) creates the GL context + accum texture */
int pt_create(pt_ctx **out, int device);
void pt_destroy(pt_ctx *ctx);
const char *pt_last_error(const pt_ctx *ctx);            /* ctx may be NULL: last pt_create error */
/* accumulation image RGBA32F W x H, zero-filled (glTexImage2D GL_RGBA32F, src/Renderer.cpp:117-128).
 * Like glDispatchCompute(W/8, H/8, 1) (src/Main.cpp:148) only the (W/8)*8 x (H/8)*8 pixels are traced. */
int pt_resize(pt_ctx *ctx, int width, int height);
/* Use caller-owned device memory (W*H*4 floats) as the accumulation image, e.g. a torch tensor that
 * an RCCL reduce will read.  NULL returns to the internal buffer.  (glBindImageTexture, src/Renderer.cpp:128) */
int pt_set_accum_buffer(pt_ctx *ctx, void *device_rgba32f, size_t bytes);
/* hipStream_t to enqueue on (NULL = the context's own stream). */
int pt_set_stream(pt_ctx *ctx, void *hip_stream);

/* --- scene: Mesh::Mesh binds Triangle[] to SSBO 1 and Node[] to SSBO 2 (src/Object.cpp:97-110);
 *     only nodes[0] (root AABB) is read by the shader (rt.comp:213).  Device layout is SoA. */
int pt_upload_mesh(pt_ctx *ctx, const PtTriangle *tris, int n_tris, const PtNode *nodes, int n_nodes);
/* Scene::SetupSSBOs / UpdateSSBOs: materials -> SSBO 3, spheres -> 4, triangles -> 5 (src/Object.cpp:3-41) */
int pt_upload_scene(pt_ctx *ctx, const PtMaterial *materials, int n_materials, const PtSphere *spheres, int n_spheres,
                    const PtTriangle *tris, int n_tris);

/* --- uniforms */
/* ComputeProgram::SetUniformCamera (src/Shader.cpp:72-76, called src/Main.cpp:146):
 * cam.position and cam.inverseView (column-major, GL_FALSE transpose) */
int pt_set_camera(pt_ctx *ctx, const float position[3], const float inverse_view[16]);
/* SetUniform1i("debugNormal") (src/Main.cpp:128-130; rt.comp:21,23,274) */
int pt_set_debug_normal(pt_ctx *ctx, int enabled);
/* the literal 3 at rt.comp:23 (12 in the legacy rt.frag:337); BASELINE configs use 4/8/16 */
int pt_set_max_bounces(pt_ctx *ctx, int max_bounces);

/* The literals in which the reference's legacy copy of the kernel, res/shaders/rt.frag, differs from res/shaders/rt.comp
 * (SURVEY App. B).  They are arguments of the same kernels (wave-uniform, one code path), the rt.comp values being the
 * defaults of a new context.  Both presets are pinned bit-exact against the reference's own shader run under llvmpipe
 * (tests/golden/radiance_128x72.npz: rt.comp, radiance_frag_128x72.npz: rt.frag); other combinations are checked against
 * the CPU oracle only.  PI / TWO_PI are not here: 6.28318530 (rt.frag:4) and 6.28318530718 (rt.comp:8) are the same float. */
typedef struct PtShadingParams {
    float hit_limit;              /* rt.comp:10,146,173: 1e-5 | rt.frag:131,166: 0 (t > 0.0; t0 > 0.0 || t1 > -0.0);  0 .. 1e-3 */
    float origin_offset;          /* rt.comp:305: 0.0002      | rt.frag:303: 0.00015 */
    float jitter_scale;           /* rt.comp:332: 0.00025     | rt.frag:332: 0.0005 */
    float sky_color[3];           /* rt.comp:25: (1, 1, 1)    | rt.frag:253: (0.92, 0.8, 0.85); also multiplies rayColor at the miss */
    float sky_intensity;          /* rt.comp:26: 0.5          | rt.frag:254: 0.8 */
    int32_t fresnel_n1_one;       /* rt.comp:283: FresnelReflectAmount(n1 = ior, ...) | rt.frag:283: n1 = 1.0 */
    int32_t two_hemisphere_draws; /* rt.comp:287: one RandomInHemisphere, used twice | rt.frag:290,295: two draws (14 per bounce) */
    int32_t pixel_center;         /* rt.comp:12: integer texel coordinates | rt.frag:14: gl_FragCoord.xy = (x + 0.5, y + 0.5) */
    int32_t sqrt_output;          /* rt.comp:322: incoming / bounces | rt.frag:320: sqrt(incoming / bounces) */
    int32_t trace_mesh;           /* rt.comp:213-218: mesh behind the root-AABB gate | rt.frag:207-216: both mesh loops commented out */
    int32_t sphere_sum_xyz;       /* sphere a, c: rt.comp:130-132 pow(length(v),2), which the reference's compiler sums (z*z + y*y) + x*x |
                                   * rt.frag:89-91,122-124 LengthSquared(v), written out as (x*x + y*y) + z*z */
} PtShadingParams;
/* preset 0 = rt.comp (what pt_create sets), 1 = rt.frag (bounces are separate: pt_set_max_bounces(ctx, 12), rt.frag:337) */
int pt_shading_preset(PtShadingParams *out, int preset);
/* Takes effect at the next render call; resets nothing else (the accumulation image is the caller's to clear).  Non-finite
 * values, a negative limit / offset / scale, hit_limit > 1e-3 and flags other than 0 / 1 return PT_E_INVALID. */
int pt_set_shading_params(pt_ctx *ctx, const PtShadingParams *params);
int pt_get_shading_params(pt_ctx *ctx, PtShadingParams *out);

/* --- tracing */
/* One accumulation pass == SetUniform1i("currAccumPass", pass) + glDispatchCompute + glMemoryBarrier
 * (src/Main.cpp:144-150): accum = mix(accum, radiance(pass), 1/pass) (rt.comp:337-339). */
int pt_render_pass(pt_ctx *ctx, int pass);
/* Passes first_pass .. first_pass+n_passes-1: all samples are traced in one launch (per workspace batch)
 * into a per-pass radiance workspace, then folded into the image in pass order; bit-identical to
 * calling pt_render_pass for each pass. */
int pt_render(pt_ctx *ctx, int first_pass, int n_passes);
/* pt_render for callers that cannot allocate (no counterpart in the reference): code under HIP graph capture (tested:
 * tests/test_zz_no_alloc.py captures a call, instantiates the graph and replays it), tight serving loops, harnesses that
 * forbid hipMalloc.  Same arguments, same image bit for bit, but it calls no hipMalloc, hipFree, event/stream creation or
 * function-attribute setter: the work-item queue, the tile schedule and the per-pass radiance workspace live in a 64 MiB array
 * that is part of the library's code object (one copy per GPU, held until the process ends).  Limits: at most 65,536 tiles
 * of 64 pixels (1920x1080 has 32,400); larger images return PT_E_LIMIT.  When the array holds fewer passes of the image than
 * a launch is worth, the passes are taken in short launches that keep the running mean in registers (same result).  The
 * scene must be uploaded before (uploads allocate).  All contexts on one GPU share that GPU's array: direct calls are ordered
 * on the device by the library (one event per GPU: a call's first launch waits for the previous call's last, whatever context
 * or stream made it, so no host synchronisation is needed between them).  A call recorded under stream capture carries no
 * such event (an event from outside a capture cannot be waited for inside it): do not run two such graphs, or a graph and a
 * direct call, on one GPU at the same time.  It records no timing events: pt_last_render_ms afterwards returns PT_E_STATE. */
int pt_render_static(pt_ctx *ctx, int first_pass, int n_passes);
/* Multi-GPU sample split (no reference counterpart; BASELINE.json north_star): adds the plain sum of
 * the 1-spp radiances of the given passes to the accumulation image (alpha += n_passes) so that
 * per-GPU buffers can be summed by an RCCL reduce; pt_scale_accum(1/spp) then yields the mean. */
int pt_render_sum(pt_ctx *ctx, int first_pass, int n_passes);
int pt_scale_accum(pt_ctx *ctx, float scale, int set_alpha_one);
/* `currAccumPass = 0` on camera movement + glClear (src/Main.cpp:110-117): zero the accumulation image. */
int pt_reset(pt_ctx *ctx);
int pt_synchronize(pt_ctx *ctx);

/* --- results */
/* the display pass samples accumTexture (compute_accum.frag:9-11, src/Main.cpp:154-156);
 * here: copy the RGBA32F image to host, row 0 = bottom row (GL texel origin) */
int pt_read_accum(pt_ctx *ctx, float *rgba, size_t bytes);
void *pt_accum_device_ptr(pt_ctx *ctx);
/* Test-only: primary-ray closest hit of `pass` per pixel, canonical ids (mesh i | n_mesh+i |
 * n_mesh+n_tris+i | -1 = miss), hit distance, normalized ray direction (3 floats/pixel) and the rng
 * state after the jitter draws.  Any output pointer may be NULL.  (CalculateRay, rt.comp:204-232) */
int pt_read_primary_hits(pt_ctx *ctx, int pass, int32_t *ids, float *t, float *dirs, uint32_t *rng_states);

/* --- multi-GPU sample split without Python (no reference counterpart; BASELINE.json north_star; SURVEY 8b last row).
 * One process (or thread) per GPU, each with its own context holding the full scene.  RCCL (librccl.so, loaded with
 * dlopen on first use: the library itself does not link against it) carries ONE reduce of the RGBA32F sum images onto
 * rank 0 per render, over xGMI inside a node.  path_tracer_amd/distributed.py is the same thing over torch.distributed. */
#define PT_GROUP_ID_BYTES 128
/* rank 0 creates the id and hands it to the other ranks by any means (file, pipe, MPI) */
int pt_group_unique_id(char id[PT_GROUP_ID_BYTES]);
/* collective over all ranks; binds the communicator to the context's device */
int pt_group_init(pt_ctx *ctx, const char id[PT_GROUP_ID_BYTES], int rank, int world);
/* collective: rank r traces its contiguous share of passes first_pass .. first_pass+total_spp-1 as a plain sum
 * (pt_render_sum semantics, accumulation image zeroed first), the sums meet on rank 0, rank 0 scales by 1/total_spp.
 * Afterwards rank 0's image is the mean of all samples (alpha 1); the other ranks' images hold their partial sums.
 * Two small all-reduces frame the render: before it the ranks compare their arguments and local state, so that a rank with
 * a missing scene or a different size makes EVERY rank return an error instead of leaving the others in the collective;
 * after the reduce they share their status again, so that a rank whose trace failed makes every rank -- rank 0 above all --
 * return PT_E_STATE instead of a darker image (rank 0 also checks the sample count the sum carries in alpha).  The call
 * returns when the reduce has finished on this rank's stream (it synchronises once, to read that status). */
int pt_group_render(pt_ctx *ctx, int first_pass, int total_spp);
int pt_group_finalize(pt_ctx *ctx);

/* --- measurement */
int pt_get_counters(pt_ctx *ctx, PtCounters *out);
int pt_reset_counters(pt_ctx *ctx);
/* Device time (ms, HIP events on the context's stream) spent in the dominant kernel (pt_trace_items) by
 * the most recent pt_render / pt_render_pass / pt_render_sum call, summed over its launches (one per
 * workspace batch; the count goes to *n_launches if not NULL); waits for them to finish. */
int pt_last_render_ms(pt_ctx *ctx, float *ms, int *n_launches);
/* Tuning knobs (0 = keep): passes per work item of the mesh tiles (default 4), the byte limit of the per-pass radiance
 * workspace [passes][H][W] x 12 B (RGB float) that bounds how many passes one launch covers (default 4 GiB), and the
 * number of persistent single-wave workgroups launched per CU (default 12 = 3 waves/SIMD), and the kernel
 * variant: 1 = brute force, mesh triangles resident in VGPRs (any mesh size), 2 / 3 / 4 = brute force, resident in a
 * workgroup-shared LDS image at 4 / 5 / 6 waves per SIMD (meshes of at most 704 triangles; default for those),
 * 5 = BVH traversal, one ray per lane, BVH in global memory (any mesh size), 6 = the same with the BVH staged into
 * LDS (default): two 512-thread workgroups per CU with a copy each while trees + stacks fit in 80 KB, else one
 * 1024-thread workgroup per CU sharing one copy (up to 158 KB), else 5; see DESIGN.md for its exactness. */
int pt_set_tuning(pt_ctx *ctx, int passes_per_item, size_t workspace_limit_bytes, int waves_per_cu, int variant);
/* Scheduling options of the traversal kernels, per context, by name (experiments and tests; none changes a result, and
 * the library reads no environment variable).  value 0 = the built-in choice for the launch at hand where noted.
 *   "park_high" (0 | 1..64)  traverse when this many lanes of a wave are parked on the mesh
 *   "park_wait" (0 | n)      ... or after this many iterations of waiting
 *   "park_exit" (0 | 1..8)   leave the traversal phase with this many eighths of its rays unfinished
 *   "refill_min" (0 | 1..64) hand out new pixels once this many lanes are idle
 *   "ppi1_mul", "ppi2_mul"   passes per work item of scene-only / sky tiles, as multiples of the mesh tiles' (4, 8)
 *   "no_fused" (0 | 1)       1: short launches also go through the per-pass workspace + pt_accumulate
 *   "wgs_per_cu" (0 | 1..8)  workgroups per CU of the LDS-resident traversal kernel
 *   "lds_block" (0|512|1024) force its workgroup size (falls back to variant 5 if the trees do not fit that way)
 *   "verify" (0 | n)         exactness self-check: every n-th pt_render / pt_render_pass / pt_render_sum call that would run a
 *                            traversal kernel is done twice from the same starting image -- by the brute-force kernel (every
 *                            mesh triangle behind the root-AABB gate, like rt.comp:213-218: exact by construction) and by the
 *                            traversal kernel -- and the two images are compared bit for bit on the device.  A difference
 *                            returns PT_E_STATE (the text names the number of pixels) and leaves the brute-force image in
 *                            place.  A checked call costs the brute-force render on top (config 2: about 4x), synchronises,
 *                            and counts both renders in pt_get_counters.  pt_render_static and pt_group_render are not checked.
 *   "verify_inject" (0 | 1)  test hook: 1 uploads the mesh tree with the root's child boxes shrunk to half their size, a
 *                            deliberately wrong tree that "verify" must catch
 * Unknown names and values out of range return PT_E_INVALID. */
int pt_set_option(pt_ctx *ctx, const char *name, int value);
/* Host-only (no HIP device needed; for tests and tools): build the library's BVH over `tris` exactly as pt_upload_mesh
 * does and verify it -- every triangle in exactly one leaf, every leaf of 1..8 triangles, every child box containing its
 * whole subtree, depth within the traversal stack.  Outputs (any may be NULL): inner nodes, depth, largest leaf, SAH cost
 * relative to the root box.  Returns PT_OK, PT_E_INVALID (bad arguments / non-finite coordinates, which pt_upload_mesh
 * answers by falling back to brute force) or PT_E_STATE (a check failed: a bug). */
int pt_bvh_selftest(const PtTriangle *tris, int n_tris, int *n_nodes, int *depth, int *max_leaf, float *sah_cost);
/* Host-only (no counterpart in the reference; for tests and tools): how pt_upload_mesh sorts the triangles of a mesh before it
 * builds the tree.  The float evaluation of rt.comp:153-173 reports hits outside a triangle whose edges at p1
 * are nearly parallel; the tree's box of every triangle is grown by 5e-4 x (longest edge) / sin(angle at p1), triangles for
 * which that exceeds four times the mesh's extent get leaves that every ray behind the root-AABB gate reaches (`listed`; more
 * than 64 of them and than half of the mesh, and the mesh is traced by the brute-force kernels), and triangles with a zero edge at p1, which the shader
 * can never hit, are left out (`never`).  max_growth: largest growth among the triangles in the tree.  Any output may be NULL. */
int pt_classify_mesh(const PtTriangle *tris, int n_tris, int *in_tree, int *listed, int *never, float *max_growth);
/* name of the library's default dominant kernel (for profiles/) */
const char *pt_kernel_name(void);
/* symbol (with template arguments) of the trace kernel the context's most recent render launched; "" before the first */
const char *pt_last_kernel_name(const pt_ctx *ctx);
int pt_version(void);

#ifdef __cplusplus
}
#endif
#endif
