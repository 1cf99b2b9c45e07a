/*
 * pt_oracle — CPU restatement of the reference's per-pixel path-tracing loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under path_tracer_amd/ (the product) may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker.  Parity status: PINNED — every function below is checked bit-for-bit against the
 * reference's own res/shaders/rt.comp executed under Mesa llvmpipe (oracle/llvmpipe_runner.c),
 * through the fixtures in tests/golden/ (generator: oracle/gen_goldens.py).
 *
 * Arithmetic model (re-verified by oracle/probes/arith_probe.comp):
 * float32, one rounding per mul/add (no contraction), correctly rounded / and sqrt,
 * dot(a,b) = (az*bz + ay*by) + ax*bx, llvmpipe's polynomial log()/cos() reproduced exactly.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* std430 records, byte-identical to the reference's host structs (src/Object.h:8-75) */
typedef struct { float baseColor[4], specularColor[4], emissionColor[4];
                 float smoothness, specularSmoothness, emissionStrength, ior, refractionAmount, specularChance;
                 int32_t pad[2]; } OrMaterial;                                   /* 80 B  rt.comp:33-44 */
typedef struct { float position[3]; float radius; uint32_t materialIndex; int32_t pad[3]; } OrSphere;     /* 32 B  rt.comp:45 */
typedef struct { float p1[4], p2[4], p3[4]; uint32_t materialIndex; int32_t pad[3]; } OrTriangle;         /* 64 B  rt.comp:46 */
typedef struct { float boundsMin[4], boundsMax[4]; int32_t triIndex, numTris, childrenIndex, pad; } OrNode; /* 48 B  rt.comp:48 */

typedef struct {
    const OrTriangle *mesh_tris; int32_t n_mesh_tris;   /* SSBO 1  rt.comp:52-54 */
    const OrNode *nodes;         int32_t n_nodes;       /* SSBO 2  rt.comp:55-57 (only nodes[0] is read, rt.comp:213) */
    const OrMaterial *materials; int32_t n_materials;   /* SSBO 3  rt.comp:58-60 */
    const OrSphere *spheres;     int32_t n_spheres;     /* SSBO 4  rt.comp:61-63 */
    const OrTriangle *tris;      int32_t n_tris;        /* SSBO 5  rt.comp:64-66 */
} OrScene;

/* The constants in which the reference's legacy copy of the kernel, res/shaders/rt.frag, differs from rt.comp.
 * Parity status of the rt.frag values: PINNED by fixtures of rt.frag itself run under llvmpipe (llvmpipe_runner --frag,
 * tests/golden/radiance_frag_128x72.npz).  PI / TWO_PI are not here: 6.28318530 (rt.frag:4) and 6.28318530718 (rt.comp:8)
 * are the same float32. */
typedef struct {
    float hit_limit;              /* rt.comp:10,146,173: 1e-5 | rt.frag:131,166: 0 (t > 0.0; t0 > 0.0 || t1 > -0.0) */
    float origin_offset;          /* rt.comp:305: 0.0002     | rt.frag:303: 0.00015 */
    float jitter_scale;           /* rt.comp:332: 0.00025    | rt.frag:332: 0.0005 */
    float sky_color[3];           /* rt.comp:25: (1,1,1)     | rt.frag:253: (0.92, 0.8, 0.85); also multiplies rayColor at the miss */
    float sky_intensity;          /* rt.comp:26: 0.5         | rt.frag:254: 0.8 */
    int32_t fresnel_n1_one;       /* rt.comp:283: n1 = ior (1/ior on a front face) | rt.frag:283: n1 = 1.0 */
    int32_t two_hemisphere_draws; /* rt.comp:287: one RandomInHemisphere, used twice | rt.frag:290,295: two draws (14 per bounce) */
    int32_t pixel_center;         /* rt.comp:12: integer texel coordinates | rt.frag:14: gl_FragCoord (x + 0.5, y + 0.5) */
    int32_t sqrt_output;          /* rt.comp:322: incoming / bounces | rt.frag:320: sqrt(incoming / bounces) */
    int32_t trace_mesh;           /* rt.comp:213-218: mesh behind the root-AABB gate | rt.frag:207-216: commented out */
    int32_t sphere_sum_xyz;       /* sphere a, c: rt.comp:130-132 pow(length(v),2) = dot(v,v), summed (z*z + y*y) + x*x by llvmpipe |
                                   * rt.frag:89-91,122-124 LengthSquared(v), written out as (x*x + y*y) + z*z */
} OrParams;

typedef struct {
    int32_t width, height;
    float cam_pos[3];        /* uniform cam.position     rt.comp:30 */
    float inv_view[16];      /* uniform cam.inverseView  rt.comp:31, column-major */
    int32_t max_bounces;     /* literal 3 at rt.comp:23; runtime here (BASELINE configs use 4/8/16) */
    int32_t debug_normal;    /* uniform debugNormal      rt.comp:21 */
    OrParams prm;            /* fill with or_params_preset(): 0 = rt.comp (the live kernel), 1 = rt.frag (legacy) */
} OrFrame;

/* per-bounce record for divergence debugging (tests/golden per-pixel traces) */
typedef struct { float origin[3], dir[3]; int32_t prim_id; float t; uint32_t rng_state; } OrBounce;

void or_params_preset(OrParams *p, int32_t preset);
/* rt.comp:68-73 */ uint32_t or_next_random(uint32_t *state);
/* rt.comp:75-77 */ float or_random_value(uint32_t *state);
/* llvmpipe log()/cos() as lowered for rt.comp:80-82 (gallivm polynomial forms, fused mads) */
float or_logf(float x);
float or_cosf(float x);
/* rt.comp:79-83 */ float or_random_value_nd(uint32_t *state);
/* rt.comp:326 (+ :12-17) */ uint32_t or_pixel_index(int32_t x, int32_t y, int32_t w, int32_t h);
/* rt.comp:327-333: seed, jitter, primary ray; returns rng state after the 6 jitter draws */
uint32_t or_primary_ray(const OrFrame *f, int32_t x, int32_t y, int32_t pass, float origin[3], float dir[3]);
/* rt.comp:204-232: canonical prim id (mesh i | n_mesh+i | n_mesh+n_tris+i | -1), hit distance */
int32_t or_closest_hit(const OrScene *s, const float o[3], const float d[3], float *t_out, int32_t *front_face, float normal[3]);
/* rt.comp:256-323 for one pixel; trace (optional, may be NULL) receives up to max_bounces records */
void or_ray_trace(const OrScene *s, const OrFrame *f, float o[3], float d[3], uint32_t *state, float rgb[3], OrBounce *trace, int32_t *n_trace);
/* rt.comp:325-340 over the whole image: one accumulation pass, accum is RGBA32F, row 0 = bottom */
void or_render_pass(const OrScene *s, const OrFrame *f, int32_t pass, float *accum_rgba);
/* passes first..first+n-1 in order */
void or_render(const OrScene *s, const OrFrame *f, int32_t first_pass, int32_t n_passes, float *accum_rgba);
/* same running mean restricted to a window of the full frame (full-size spot checks) */
void or_render_region(const OrScene *s, const OrFrame *f, int32_t first_pass, int32_t n_passes, int32_t x0, int32_t y0, int32_t w, int32_t h, float *out);
/* 1-spp radiance of one pass without accumulation (rgb, 3 floats per pixel) */
void or_radiance_pass(const OrScene *s, const OrFrame *f, int32_t pass, float *rgb);
/* primary-ray closest hit per pixel: ids (int32), t (float), dirs (3 floats), rng state after jitter; any may be NULL */
void or_primary_hits(const OrScene *s, const OrFrame *f, int32_t pass, int32_t *ids, float *t, float *dirs, uint32_t *states);
/* counts executed path segments / mesh-gate passes for roofline accounting (bench only) */
void or_count_work(const OrScene *s, const OrFrame *f, int32_t first_pass, int32_t n_passes, uint64_t counts[4]);
int32_t or_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
