/*
 * pt_oracle.c — CPU restatement of res/shaders/rt.comp (reference @ /root/reference).
 * TEST INFRASTRUCTURE ONLY; see pt_oracle.h.  Compile with -ffp-contract=off (oracle/Makefile):
 * every `*`, `+`, `-` below is one IEEE float32 rounding, exactly as llvmpipe evaluates the
 * shader.  fmaf() appears only where llvmpipe itself emits a fused mad
 * (inside its log/cos polynomials).
 */
#include "pt_oracle.h"
#include <math.h>
#include <string.h>
#include <omp.h>

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* dot: llvmpipe sums z,y,x (App. A.0) */
static inline float dot(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
static inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* normalize(v) = v * (1/sqrt(dot(v,v))) (App. A.0) */
static inline v3 normalize(v3 v) { float r = 1.0f / sqrtf(dot(v, v)); return scale(v, r); }
/* mix with a dynamic t lowers to x + t*(y-x) (App. A.0) */
static inline float mixf(float x, float y, float t) { return x + t * (y - x); }
static inline v3 mix3(v3 x, v3 y, float t) { return V(mixf(x.x, y.x, t), mixf(x.y, y.y, t), mixf(x.z, y.z, t)); }
/* x86 minps/maxps semantics (second operand on NaN), which is what llvmpipe emits */
static inline float minps(float a, float b) { return a < b ? a : b; }
static inline float maxps(float a, float b) { return a > b ? a : b; }

#define PT_INFINITY 10000000.0f /* rt.comp:9  */
#define TWO_PI 6.28318530718f   /* rt.comp:8  */

void or_params_preset(OrParams *p, int32_t preset)
{
    memset(p, 0, sizeof *p);
    if (preset == 0) {            /* res/shaders/rt.comp */
        p->hit_limit = 0.00001f; p->origin_offset = 0.0002f; p->jitter_scale = 0.00025f;
        p->sky_color[0] = p->sky_color[1] = p->sky_color[2] = 1.0f; p->sky_intensity = 0.5f;
        p->trace_mesh = 1;
    } else {                      /* res/shaders/rt.frag */
        p->hit_limit = 0.0f; p->origin_offset = 0.00015f; p->jitter_scale = 0.0005f;
        p->sky_color[0] = 0.92f; p->sky_color[1] = 0.8f; p->sky_color[2] = 0.85f; p->sky_intensity = 0.8f;
        p->fresnel_n1_one = 1; p->two_hemisphere_draws = 1; p->pixel_center = 1; p->sqrt_output = 1; p->trace_mesh = 0; p->sphere_sum_xyz = 1;
    }
}

/* ---- RNG (rt.comp:68-77) ---- */
uint32_t or_next_random(uint32_t *state)
{
    uint32_t s = *state * 747796405u + 2891336453u;
    *state = s;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (r >> 22) ^ r;
}

float or_random_value(uint32_t *state)
{
    /* literal 4294967295.0 is not representable in float32: it is 2^32 (rt.comp:76) */
    return (float)or_next_random(state) / 4294967296.0f;
}

/* ---- llvmpipe's log(): log2 polynomial (gallivm lp_build_log2_approx, degree-4 minimax in
 * y^2 with y = (m-1)/(m+1), even/odd Horner with fused mads) times ln2.  Verified bit-exact
 * against llvmpipe on 2^20 inputs in (0,1) and 2^20 RNG outputs (oracle/probes). ---- */
float or_logf(float x)
{
    union { float f; uint32_t u; } c; c.f = x;
    int32_t e = (int32_t)((c.u & 0x7f800000u) >> 23) - 127;
    float logexp = (float)e;
    union { float f; uint32_t u; } m; m.u = (c.u & 0x007fffffu) | 0x3f800000u;
    float y = (m.f - 1.0f) / (m.f + 1.0f);
    float z = y * y;
    const float c0 = (float)2.88539009343309178325, c1 = (float)0.961791550404184197881, c2 = (float)0.577440339438736392009,
                c3 = (float)0.403343858251329912514, c4 = (float)0.406718052498846252698;
    float z2 = z * z;
    float even = fmaf(z2, fmaf(z2, c4, c2), c0);
    float odd = fmaf(z2, c3, c1);
    float p = fmaf(odd, z, even);
    float res = fmaf(y, p, logexp);
    if (x >= INFINITY) res = INFINITY;
    if (x == 0.0f) res = -INFINITY;
    if (x < 0.0f) res = NAN;
    return res * (float)0.69314718055994529;
}

/* ---- llvmpipe's cos(): Cephes-style range reduction by pi/4 octants, two degree-3
 * polynomials in z = x^2, fused mads where gallivm uses lp_build_fmuladd.  Verified bit-exact
 * against llvmpipe for arguments 2*pi*u, u in [0,1). ---- */
float or_cosf(float a)
{
    float x_abs = fabsf(a);
    float scale_y = x_abs * (float)1.27323954473516;
    int32_t j = (int32_t)scale_y;
    j = (j + 1) & ~1;
    float y = (float)j;
    int32_t j2 = j - 2;
    uint32_t sign_bit = ((uint32_t)(4 & ~j2)) << 29;
    int use_sin_poly = (j2 & 2) == 0;
    float x = fmaf(y, (float)-0.78515625, x_abs);
    x = fmaf(y, (float)-2.4187564849853515625e-4, x);
    x = fmaf(y, (float)-3.77489497744594108e-8, x);
    float z = x * x;
    float yc = fmaf(z, (float)2.443315711809948E-005, (float)-1.388731625493765E-003);
    yc = fmaf(yc, z, (float)4.166664568298827E-002);
    yc = yc * z;
    yc = yc * z;
    yc = yc - z * 0.5f;
    yc = yc + 1.0f;
    float ys = fmaf(z, (float)-1.9515295891E-4, (float)8.3321608736E-3);
    ys = fmaf(ys, z, (float)-1.6666654611E-1);
    ys = ys * z;
    ys = fmaf(ys, x, x);
    union { float f; uint32_t u; } r; r.f = use_sin_poly ? ys : yc;
    r.u ^= sign_bit;
    float res = minps(maxps(r.f, -1.0f), 1.0f);
    if (!(x_abs < INFINITY)) res = NAN;
    return res;
}

/* rt.comp:79-83: theta drawn first, then rho */
float or_random_value_nd(uint32_t *state)
{
    float theta = TWO_PI * or_random_value(state);
    float rho = sqrtf(-2.0f * or_logf(or_random_value(state)));
    return rho * or_cosf(theta);
}

/* rt.comp:85-90 */
static v3 random_direction(uint32_t *state)
{
    float x = or_random_value_nd(state);
    float y = or_random_value_nd(state);
    float z = or_random_value_nd(state);
    return normalize(V(x, y, z));
}

/* rt.comp:92-95 */
static v3 random_in_hemisphere(v3 n, uint32_t *state)
{
    v3 r = random_direction(state);
    return dot(r, n) > 0.0f ? r : neg(r);
}

/* rt.comp:12-13,16,326: pixelIndex = uint(vx/W*2^32 + vy/H*2^32) */
static uint32_t pixel_index_f(float fx, float fy, int32_t w, int32_t h)
{
    float W = (float)w, H = (float)h;
    float vx = fx / W, vy = fy / H;
    float v = (vx / W) * 4294967296.0f + (vy / H) * 4294967296.0f;
    return (uint32_t)v;
}
uint32_t or_pixel_index(int32_t x, int32_t y, int32_t w, int32_t h) { return pixel_index_f((float)x, (float)y, w, h); }

/* rt.comp:326-333 */
uint32_t or_primary_ray(const OrFrame *f, int32_t x, int32_t y, int32_t pass, float origin[3], float dir[3])
{
    float W = (float)f->width, H = (float)f->height;
    /* rt.comp:12 integer texel coordinates; rt.frag:14 gl_FragCoord.xy = pixel centres (x + 0.5 is exact in float32) */
    float fx = f->prm.pixel_center ? (float)x + 0.5f : (float)x, fy = f->prm.pixel_center ? (float)y + 0.5f : (float)y;
    float vx = fx / W, vy = fy / H;
    float px = ((vx - 0.5f) * W) / H;       /* rt.comp:14,17 */
    float py = vy - 0.5f;
    uint32_t state = (uint32_t)pass * pixel_index_f(fx, fy, f->width, f->height);   /* rt.comp:327 */
    /* rt.comp:332 `RandomDirection(s) * 0.00025`: llvmpipe's compiler moves the constant inside the
     * normalize, (v * 0.00025) * (1/sqrt(dot(v,v))) -- pinned by tests/golden primary dirs */
    float gx = or_random_value_nd(&state), gy = or_random_value_nd(&state), gz = or_random_value_nd(&state);
    v3 g = V(gx, gy, gz);
    float gr = 1.0f / sqrtf(dot(g, g));
    v3 jit = scale(scale(g, f->prm.jitter_scale), gr);
    float X = -px + jit.x, Y = py + jit.y;
    const float *m = f->inv_view;           /* column-major: column k = m[4k..4k+3] */
    /* mat4 * vec4(X, Y, 1, 0) lowers to (c0*X + c2) + c1*Y (App. A.0) */
    v3 d = V((m[0] * X + m[8]) + m[4] * Y, (m[1] * X + m[9]) + m[5] * Y, (m[2] * X + m[10]) + m[6] * Y);
    d = normalize(d);
    origin[0] = f->cam_pos[0]; origin[1] = f->cam_pos[1]; origin[2] = f->cam_pos[2];
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
    return state;
}

/* rt.comp:102-124 */
static int hit_aabb(const float bmin[4], const float bmax[4], v3 o, v3 d)
{
    float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    float tx1 = (bmin[0] - o.x) * ix, tx2 = (bmax[0] - o.x) * ix;
    float tmin = minps(tx1, tx2), tmax = maxps(tx1, tx2);
    float ty1 = (bmin[1] - o.y) * iy, ty2 = (bmax[1] - o.y) * iy;
    tmin = maxps(tmin, minps(ty1, ty2));
    tmax = minps(tmax, maxps(ty1, ty2));
    float tz1 = (bmin[2] - o.z) * iz, tz2 = (bmax[2] - o.z) * iz;
    tmin = maxps(tmin, minps(tz1, tz2));
    tmax = minps(tmax, maxps(tz1, tz2));
    return tmax >= tmin && tmax >= 0.0f;
}

/* rt.comp:153-180, hit test part only (normal/material are evaluated for the winner) */
static inline int hit_triangle(const OrTriangle *tri, v3 o, v3 d, float hit_limit, float *t_out, float *det_out)
{
    v3 p1 = V(tri->p1[0], tri->p1[1], tri->p1[2]);
    v3 e1 = sub(V(tri->p2[0], tri->p2[1], tri->p2[2]), p1);
    v3 e2 = sub(V(tri->p3[0], tri->p3[1], tri->p3[2]), p1);
    v3 q = cross(d, e2);
    float det = dot(e1, q);
    float inv = 1.0f / det;
    v3 s = sub(o, p1);
    float u = inv * dot(s, q);
    v3 r = cross(s, e1);
    float v = inv * dot(d, r);
    float t = inv * dot(e2, r);
    /* `!(det < 0.0 && det > -0.0)` (rt.comp:173) is always true */
    /* rt.frag:151-166 returns early instead: the same predicate, NaN u / v included, with t > 0.0 */
    int has = t > hit_limit && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f);
    *t_out = t; *det_out = det;
    return has;
}

/* rt.comp:126-150; pow(length(v),2) folds to dot(v,v), pow(r,2) to r*r (App. A.0) */
static inline float length_squared(v3 v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }     /* rt.frag:89-91 */
static inline int hit_sphere(const OrSphere *sp, v3 o, v3 d, float hit_limit, int sum_xyz, float *t_out, int *front_out)
{
    v3 oc = sub(o, V(sp->position[0], sp->position[1], sp->position[2]));
    float a = sum_xyz ? length_squared(d) : dot(d, d);
    float hb = dot(oc, d);
    float c = (sum_xyz ? length_squared(oc) : dot(oc, oc)) - sp->radius * sp->radius;
    float disc = hb * hb - a * c;
    float sq = sqrtf(disc);
    float t0 = (-hb - sq) / a;
    float t1 = (-hb + sq) / a;
    int front = t0 > 0.0f;
    *t_out = front ? t0 : t1;
    *front_out = front;
    return t0 > hit_limit || t1 > -hit_limit;            /* rt.frag:131: t0 > 0.0 || t1 > -0.0 */
}

typedef struct { int32_t id; float t; int front; v3 normal; v3 point; uint32_t mat; int gate; } Hit;

/* rt.comp:204-232 */
static Hit calculate_ray(const OrScene *s, const OrParams *prm, v3 o, v3 d)
{
    const float hit_limit = prm->hit_limit;
    Hit h; memset(&h, 0, sizeof h);
    h.id = -1; h.t = PT_INFINITY;
    const OrTriangle *win_tri = 0; const OrSphere *win_sph = 0; float win_det = 0.0f;
    h.gate = prm->trace_mesh && s->n_nodes > 0 && hit_aabb(s->nodes[0].boundsMin, s->nodes[0].boundsMax, o, d);
    if (h.gate) {
        for (int i = 0; i < s->n_mesh_tris; i++) {
            float t, det;
            if (hit_triangle(&s->mesh_tris[i], o, d, hit_limit, &t, &det) && t < h.t) { h.t = t; h.id = i; win_tri = &s->mesh_tris[i]; win_det = det; }
        }
    }
    for (int i = 0; i < s->n_tris; i++) {
        float t, det;
        if (hit_triangle(&s->tris[i], o, d, hit_limit, &t, &det) && t < h.t) { h.t = t; h.id = s->n_mesh_tris + i; win_tri = &s->tris[i]; win_det = det; }
    }
    for (int i = 0; i < s->n_spheres; i++) {
        float t; int front;
        if (hit_sphere(&s->spheres[i], o, d, hit_limit, prm->sphere_sum_xyz, &t, &front) && t < h.t) { h.t = t; h.id = s->n_mesh_tris + s->n_tris + i; win_tri = 0; win_sph = &s->spheres[i]; h.front = front; }
    }
    if (h.id < 0) return h;
    h.point = add(o, scale(d, h.t));
    if (win_tri) {
        v3 p1 = V(win_tri->p1[0], win_tri->p1[1], win_tri->p1[2]);
        v3 e1 = sub(V(win_tri->p2[0], win_tri->p2[1], win_tri->p2[2]), p1);
        v3 e2 = sub(V(win_tri->p3[0], win_tri->p3[1], win_tri->p3[2]), p1);
        v3 n = normalize(cross(e2, e1));
        h.front = !(win_det < 0.0f);                    /* rt.comp:174 */
        h.normal = win_det < 0.0f ? n : neg(n);         /* rt.comp:177 */
        h.mat = win_tri->materialIndex;
    } else {
        v3 n = normalize(sub(h.point, V(win_sph->position[0], win_sph->position[1], win_sph->position[2])));
        h.normal = h.front ? n : neg(n);                /* rt.comp:147 */
        h.mat = win_sph->materialIndex;
    }
    return h;
}

int32_t or_closest_hit(const OrScene *s, const float o[3], const float d[3], float *t_out, int32_t *front_face, float normal[3])
{
    OrParams prm; or_params_preset(&prm, 0);
    Hit h = calculate_ray(s, &prm, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
    if (t_out) *t_out = h.t;
    if (front_face) *front_face = h.front;
    if (normal) { normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z; }
    return h.id;
}

/* rt.comp:235-254 */
static float fresnel_reflect_amount(float n1, float n2, v3 normal, v3 incident, float reflectivity)
{
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cosX = -dot(normal, incident);
    if (n1 > n2) {
        float n = n1 / n2;
        float sinT2 = (n * n) * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = sqrtf(1.0f - sinT2);
    }
    float x = 1.0f - cosX;
    float ret = r0 + (((((1.0f - r0) * x) * x) * x) * x) * x;
    return reflectivity + (1.0f - reflectivity) * ret;
}

static inline v3 reflect3(v3 i, v3 n) { float k = 2.0f * dot(n, i); return sub(i, scale(n, k)); }
static inline v3 refract3(v3 i, v3 n, float eta)
{
    float d = dot(n, i);
    /* Mesa's refract builtin: k = 1 - eta*(eta*(1 - d*d)) (a literal eta folds to (eta*eta)*..., a dynamic one does not) */
    float k = 1.0f - eta * (eta * (1.0f - d * d));
    if (k < 0.0f) return V(0.0f, 0.0f, 0.0f);
    float c = eta * d + sqrtf(k);
    return sub(scale(i, eta), scale(n, c));
}

/* rt.comp:256-323 */
static v3 ray_trace(const OrScene *s, const OrFrame *f, v3 o, v3 d, uint32_t *state, OrBounce *trace, int32_t *n_trace, uint64_t *counts)
{
    v3 ray_color = V(1, 1, 1), incoming = V(0, 0, 0), emitted = V(0, 0, 0);
    int max_bounces = f->debug_normal ? 1 : f->max_bounces;     /* rt.comp:23 */
    int bounces = 0;
    const OrParams *prm = &f->prm;
    /* skyColor * skyIntensity is a constant expression in both shaders: one float32 product per channel */
    const v3 sky = V(prm->sky_color[0], prm->sky_color[1], prm->sky_color[2]), sky_emit = scale(sky, prm->sky_intensity);
    for (int i = 0; i < max_bounces; i++) {
        Hit h = calculate_ray(s, prm, o, d);
        if (counts) { counts[0]++; counts[1] += h.gate; }
        if (trace) {
            OrBounce *b = &trace[*n_trace]; (*n_trace)++;
            b->origin[0] = o.x; b->origin[1] = o.y; b->origin[2] = o.z;
            b->dir[0] = d.x; b->dir[1] = d.y; b->dir[2] = d.z;
            b->prim_id = h.id; b->t = h.t; b->rng_state = *state;
        }
        if (h.id >= 0 && h.t < PT_INFINITY) {
            if (f->debug_normal) return h.normal;               /* rt.comp:274 */
            if (counts) counts[2]++;
            bounces++;
            const OrMaterial *m = &s->materials[h.mat];
            o = h.point;
            float ior = h.front ? (1.0f / m->ior) : m->ior;
            int is_spec = m->specularChance > or_random_value(state);
            float fres = fresnel_reflect_amount(prm->fresnel_n1_one ? 1.0f : ior, m->ior, h.normal, d, 1.0f - m->refractionAmount);   /* rt.comp:283 | rt.frag:283 */
            int is_refr = fres < or_random_value(state);
            v3 hemi = random_in_hemisphere(h.normal, state);
            /* rt.frag:290,295 draws a second hemisphere direction for the refracted branch (both mix() arguments are
             * evaluated, in source order under llvmpipe: pinned by the rt.frag fixtures) */
            v3 hemi2 = prm->two_hemisphere_draws ? random_in_hemisphere(h.normal, state) : hemi;
            v3 a = mix3(hemi, reflect3(d, h.normal), is_spec ? m->specularSmoothness : m->smoothness);
            v3 b = mix3(neg(hemi2), refract3(d, h.normal, ior), m->smoothness);
            d = normalize(is_refr ? b : a);
            o = add(o, scale(d, prm->origin_offset));
            v3 base = V(m->baseColor[0], m->baseColor[1], m->baseColor[2]);
            v3 spec = V(m->specularColor[0], m->specularColor[1], m->specularColor[2]);
            emitted = add(emitted, scale(V(m->emissionColor[0], m->emissionColor[1], m->emissionColor[2]), m->emissionStrength));
            ray_color = mul(ray_color, is_refr ? base : (is_spec ? spec : base));
            incoming = add(incoming, mul(emitted, ray_color));
        } else {
            bounces++;
            emitted = add(emitted, sky_emit);                   /* skyColor * skyIntensity, rt.comp:315 */
            ray_color = mul(ray_color, sky);                    /* rt.comp:316: * vec3(1), exact */
            incoming = add(incoming, mul(emitted, ray_color));
            break;
        }
    }
    float nb = (float)bounces;
    v3 c = V(incoming.x / nb, incoming.y / nb, incoming.z / nb);
    if (prm->sqrt_output) c = V(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z));      /* rt.frag:320 */
    return c;
}

void or_ray_trace(const OrScene *s, const OrFrame *f, float o[3], float d[3], uint32_t *state, float rgb[3], OrBounce *trace, int32_t *n_trace)
{
    int32_t n = 0;
    v3 c = ray_trace(s, f, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), state, trace, &n, 0);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    if (n_trace) *n_trace = n;
}

static v3 sample_pixel(const OrScene *s, const OrFrame *f, int x, int y, int pass, uint64_t *counts)
{
    float o[3], d[3];
    uint32_t state = or_primary_ray(f, x, y, pass, o, d);
    int32_t n = 0;
    return ray_trace(s, f, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), &state, 0, &n, counts);
}

void or_radiance_pass(const OrScene *s, const OrFrame *f, int32_t pass, float *rgb)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < f->height; y++)
        for (int x = 0; x < f->width; x++) {
            v3 c = sample_pixel(s, f, x, y, pass, 0);
            float *p = rgb + 3 * ((size_t)y * f->width + x);
            p[0] = c.x; p[1] = c.y; p[2] = c.z;
        }
}

/* rt.comp:337-339: final = mix(prev, colour, 1.0/pass) -> prev + t*(colour - prev); alpha = 1 */
void or_render_pass(const OrScene *s, const OrFrame *f, int32_t pass, float *accum)
{
    float t = 1.0f / (float)pass;
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < f->height; y++)
        for (int x = 0; x < f->width; x++) {
            v3 c = sample_pixel(s, f, x, y, pass, 0);
            float *p = accum + 4 * ((size_t)y * f->width + x);
            p[0] = mixf(p[0], c.x, t); p[1] = mixf(p[1], c.y, t); p[2] = mixf(p[2], c.z, t); p[3] = 1.0f;
        }
}

void or_render(const OrScene *s, const OrFrame *f, int32_t first_pass, int32_t n_passes, float *accum)
{
    for (int p = 0; p < n_passes; p++) or_render_pass(s, f, first_pass + p, accum);
}

/* running mean of passes first..first+n-1 for the w x h window at (x0,y0) of the full frame; out is (h,w,4), zero-based on entry */
void or_render_region(const OrScene *s, const OrFrame *f, int32_t first_pass, int32_t n_passes, int32_t x0, int32_t y0, int32_t w, int32_t h, float *out)
{
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float *p = out + 4 * ((size_t)y * w + x);
            for (int k = 0; k < n_passes; k++) {
                int pass = first_pass + k;
                float t = 1.0f / (float)pass;
                v3 c = sample_pixel(s, f, x0 + x, y0 + y, pass, 0);
                p[0] = mixf(p[0], c.x, t); p[1] = mixf(p[1], c.y, t); p[2] = mixf(p[2], c.z, t); p[3] = 1.0f;
            }
        }
}

void or_primary_hits(const OrScene *s, const OrFrame *f, int32_t pass, int32_t *ids, float *t, float *dirs, uint32_t *states)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < f->height; y++)
        for (int x = 0; x < f->width; x++) {
            float o[3], d[3];
            uint32_t st = or_primary_ray(f, x, y, pass, o, d);
            size_t i = (size_t)y * f->width + x;
            if (ids || t) {
                Hit h = calculate_ray(s, &f->prm, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
                if (ids) ids[i] = h.id;
                if (t) t[i] = h.t;
            }
            if (dirs) { dirs[3 * i] = d[0]; dirs[3 * i + 1] = d[1]; dirs[3 * i + 2] = d[2]; }
            if (states) states[i] = st;
        }
}

void or_count_work(const OrScene *s, const OrFrame *f, int32_t first_pass, int32_t n_passes, uint64_t counts[4])
{
    uint64_t c0 = 0, c1 = 0, c2 = 0;
    for (int p = 0; p < n_passes; p++) {
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : c0, c1, c2)
        for (int y = 0; y < f->height; y++)
            for (int x = 0; x < f->width; x++) {
                uint64_t c[3] = { 0, 0, 0 };
                sample_pixel(s, f, x, y, first_pass + p, c);
                c0 += c[0]; c1 += c[1]; c2 += c[2];
            }
    }
    counts[0] = c0; counts[1] = c1; counts[2] = c2;
    counts[3] = (uint64_t)f->width * f->height * n_passes;
}

int32_t or_num_threads(void) { return omp_get_max_threads(); }
