// pt_math.hpp — device-side float32 arithmetic of the path tracer, in the exact operation order of
// the reference shader as llvmpipe evaluates it (pinned by tests/golden).
// Rules: this translation unit is compiled with -ffp-contract=off, so every `*`, `+`, `-` is ONE
// IEEE rounding (no FMA contraction); `/` and sqrtf are correctly rounded
// (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default); fmaf() appears only where the
// reference's GLSL compiler itself emits a fused multiply-add (inside log()/cos()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace pt {

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
// dot(a,b) = (az*bz + ay*by) + ax*bx  -- llvmpipe's summation order
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// normalize(v) = v * (1/sqrt(dot(v,v)))
__device__ __forceinline__ float exact_rcp(float x);
// Correctly rounded sqrt in 5 instructions: s = x * v_rsq_f32(x), then one fused residual step s + (x - s*s) * (rsq/2).
// Equal to sqrtf(x) for EVERY x in [2^-64, inf) -- all 1.6e9 of them compared by tools/sqrt_exact.hip
// (profiles/sqrt_exact_mi355x.txt); anything else (0, tiny, negative, inf, NaN) takes the IEEE expansion.
__device__ __forceinline__ float exact_sqrt(float x)
{
    const float rs = __builtin_amdgcn_rsqf(x);
    const float sb = x * rs;
    float s = fmaf(fmaf(-sb, sb, x), 0.5f * rs, sb);
    if (__builtin_expect(!((__float_as_uint(x) - 0x1f800000u) < 0x60000000u), 0)) s = sqrtf(x);
    return s;
}
__device__ __forceinline__ v3 normalize(v3 v) { float r = exact_rcp(exact_sqrt(dot(v, v))); return v * r; }
// mix(x,y,t) with a dynamic t = x + t*(y-x)
__device__ __forceinline__ float mixf(float x, float y, float t) { return x + t * (y - x); }
__device__ __forceinline__ v3 mix3(v3 x, v3 y, float t) { return V(mixf(x.x, y.x, t), mixf(x.y, y.y, t), mixf(x.z, y.z, t)); }
// x86 minps / maxps (second operand when unordered), which is what the reference run emits for min()/max()
__device__ __forceinline__ float minps(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float maxps(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ v3 select3(bool c, v3 a, v3 b) { return V(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

constexpr float kInfinity = 10000000.0f;  // rt.comp:9
constexpr float kHitLimit = 0.00001f;     // rt.comp:10 (the default of Shading::hit_limit)
constexpr float kTwoPi = 6.28318530718f;  // rt.comp:8

// ---- RNG (rt.comp:68-77) ----
__device__ __forceinline__ uint32_t next_random(uint32_t &state)
{
    state = state * 747796405u + 2891336453u;
    uint32_t r = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (r >> 22) ^ r;
}
// the literal 4294967295.0 is 2^32 in float32, so this is an exact scaling of the RNE-converted integer
__device__ __forceinline__ float random_value(uint32_t &state) { return (float)next_random(state) / 4294967296.0f; }

// ---- log(): the reference's GLSL compiler (Mesa gallivm) evaluates log(x) = log2(x)*ln2 with a
// degree-4 minimax polynomial in y^2, y = (m-1)/(m+1), even/odd Horner with fused mads.
__device__ __forceinline__ float pt_logf(float x)
{
    uint32_t u = __float_as_uint(x);
    float logexp = (float)((int32_t)((u & 0x7f800000u) >> 23) - 127);
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    // y = (m - 1) / (m + 1): v_rcp_f32 + one residual correction equals the IEEE quotient for every m in [1, 2) -- all
    // 2^23 of them checked by tools/div_exact.hip (profiles/div_exact_mi355x.txt); 4 instructions instead of 12
    const float ya = m - 1.0f, yb = m + 1.0f;
    const float yr = __builtin_amdgcn_rcpf(yb);
    const float yq = ya * yr;
    float y = fmaf(fmaf(-yb, yq, ya), yr, yq);
    float z = y * y;
    const float c0 = (float)2.88539009343309178325, c1 = (float)0.961791550404184197881, c2 = (float)0.577440339438736392009,
                c3 = (float)0.403343858251329912514, c4 = (float)0.406718052498846252698;
    float z2 = z * z;
    float even = fmaf(z2, fmaf(z2, c4, c2), c0);
    float odd = fmaf(z2, c3, c1);
    float p = fmaf(odd, z, even);
    float res = fmaf(y, p, logexp);
    if (x >= __builtin_inff()) res = __builtin_inff();
    if (x == 0.0f) res = -__builtin_inff();
    if (x < 0.0f) res = __builtin_nanf("");
    return res * (float)0.69314718055994529;
}

// ---- cos(): Cephes-style octant reduction + two degree-3 polynomials in z = x^2 (same compiler).
__device__ __forceinline__ float pt_cosf(float a)
{
    float x_abs = fabsf(a);
    float scale_y = x_abs * (float)1.27323954473516;
    int32_t j = (int32_t)scale_y;
    j = (j + 1) & ~1;
    float y = (float)j;
    int32_t j2 = j - 2;
    uint32_t sign_bit = ((uint32_t)(4 & ~j2)) << 29;
    bool use_sin_poly = (j2 & 2) == 0;
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
    float r = __uint_as_float(__float_as_uint(use_sin_poly ? ys : yc) ^ sign_bit);
    float res = minps(maxps(r, -1.0f), 1.0f);
    if (!(x_abs < __builtin_inff())) res = __builtin_nanf("");
    return res;
}

// rt.comp:79-83: theta is drawn first, then rho
__device__ __forceinline__ float random_value_nd_generic(uint32_t &state)
{
    float theta = kTwoPi * random_value(state);
    float rho = exact_sqrt(-2.0f * pt_logf(random_value(state)));
    return rho * pt_cosf(theta);
}
// The two factors are functions of ONE 32-bit draw each, so their special-case handling can be cut down to what those
// 2^32 inputs need: theta = 2 pi u / 2^32 is finite and >= 0 and the polynomial result never leaves [-1, 1] (no NaN
// test, no clamp); u / 2^32 <= 1 (no +inf test in log).  tools/nd_exact.hip compares both with pt_cosf / pt_logf on every
// u (profiles/nd_exact_mi355x.txt).
__device__ __forceinline__ float nd_cos_of_draw(uint32_t r)
{
    // kTwoPi * (r / 2^32): scaling by a power of two commutes with the rounding, so one multiply by kTwoPi * 2^-32 (exact)
    const float a = (float)r * (kTwoPi * 2.3283064365386963e-10f);
    const float scale_y = a * (float)1.27323954473516;
    int32_t j = (int32_t)scale_y;
    j = (j + 1) & ~1;
    const float y = (float)j;
    const int32_t j2 = j - 2;
    const uint32_t sign_bit = ((uint32_t)(4 & ~j2)) << 29;
    const bool use_sin_poly = (j2 & 2) == 0;
    float x = fmaf(y, (float)-0.78515625, a);
    x = fmaf(y, (float)-2.4187564849853515625e-4, x);
    x = fmaf(y, (float)-3.77489497744594108e-8, x);
    const float z = x * x;
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
    return __uint_as_float(__float_as_uint(use_sin_poly ? ys : yc) ^ sign_bit);
}
__device__ __forceinline__ float nd_rho_of_draw(uint32_t r)
{
    // log(r / 2^32) from the bits of (float)r itself: same mantissa, exponent lower by 32.  r = 0 (log = -inf, rho = +inf)
    // needs no test of its own: its exponent field 0 makes the argument of the square root 220, far above the 44.4 any
    // other draw can reach, so it leaves through the range check of the short square root.
    const uint32_t u = __float_as_uint((float)r);
    const float logexp = (float)((int32_t)(u >> 23) - 159);
    const float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    const float ya = m - 1.0f, yb = m + 1.0f;
    const float yr = __builtin_amdgcn_rcpf(yb);
    const float yq = ya * yr;
    const float y = fmaf(fmaf(-yb, yq, ya), yr, yq);
    const float z = y * y;
    const float c0 = (float)2.88539009343309178325, c1 = (float)0.961791550404184197881, c2 = (float)0.577440339438736392009,
                c3 = (float)0.403343858251329912514, c4 = (float)0.406718052498846252698;
    const float z2 = z * z;
    const float even = fmaf(z2, fmaf(z2, c4, c2), c0);
    const float odd = fmaf(z2, c3, c1);
    const float res = fmaf(y, fmaf(odd, z, even), logexp);
    const float arg = -2.0f * (res * (float)0.69314718055994529);
    // exact_sqrt's short form on [2^-64, 64); r = 0 and the draws that round to u = 1 (arg = -0) take the slow path
    const float rs = __builtin_amdgcn_rsqf(arg);
    const float sb = arg * rs;
    float s = fmaf(fmaf(-sb, sb, arg), 0.5f * rs, sb);
    if (__builtin_expect(!((__float_as_uint(arg) - 0x1f800000u) < (0x42800000u - 0x1f800000u)), 0)) s = r == 0u ? __builtin_inff() : sqrtf(arg);
    return s;
}
__device__ __forceinline__ float random_value_nd(uint32_t &state)
{
    const uint32_t r_theta = next_random(state), r_rho = next_random(state);
    return nd_rho_of_draw(r_rho) * nd_cos_of_draw(r_theta);
}

// rt.comp:85-90
__device__ __forceinline__ v3 random_direction(uint32_t &state)
{
    float x = random_value_nd(state);
    float y = random_value_nd(state);
    float z = random_value_nd(state);
    return normalize(V(x, y, z));
}

// rt.comp:92-95
__device__ __forceinline__ v3 random_in_hemisphere(v3 n, uint32_t &state)
{
    v3 r = random_direction(state);
    return dot(r, n) > 0.0f ? r : -r;
}

// rt.comp:12-13,16,326; fx, fy = the texel's integer coordinates (rt.comp:12) or its centre (gl_FragCoord, rt.frag:14)
__device__ __forceinline__ uint32_t pixel_index(float fx, float fy, float W, float H)
{
    float vx = fx / W, vy = fy / H;
    float v = (vx / W) * 4294967296.0f + (vy / H) * 4294967296.0f;
    return (uint32_t)v;
}

// rt.comp:102-124
__device__ __forceinline__ bool hit_aabb(v3 bmin, v3 bmax, v3 o, v3 d)
{
    float ix = exact_rcp(d.x), iy = exact_rcp(d.y), iz = exact_rcp(d.z);      // == 1.0f / d.*, bit for bit
    float tx1 = (bmin.x - o.x) * ix, tx2 = (bmax.x - o.x) * ix;
    float tmin = minps(tx1, tx2), tmax = maxps(tx1, tx2);
    float ty1 = (bmin.y - o.y) * iy, ty2 = (bmax.y - o.y) * iy;
    tmin = maxps(tmin, minps(ty1, ty2));
    tmax = minps(tmax, maxps(ty1, ty2));
    float tz1 = (bmin.z - o.z) * iz, tz2 = (bmax.z - o.z) * iz;
    tmin = maxps(tmin, minps(tz1, tz2));
    tmax = minps(tmax, maxps(tz1, tz2));
    return tmax >= tmin && tmax >= 0.0f;
}

#ifndef PT_FAST_RCP
#define PT_FAST_RCP 1
#endif
#ifndef PT_MINMAX
#define PT_MINMAX 1
#endif
#ifndef PT_SLOAD
#define PT_SLOAD 1
#endif
// Correctly rounded 1/x for the hot loop.  v_rcp_f32 (<= 1 ulp) + one fused Newton step equals the
// IEEE quotient whenever the result is a normal number -- checked on all 2^32 bit patterns by
// tools/rcp_exact.hip.  `bad` is raised otherwise (zero, denormal, huge, inf or NaN operand); the caller
// then repeats the work with the full IEEE divide (a wave-uniform branch that is practically never taken).
__device__ __forceinline__ float fast_rcp_checked(float x, bool &bad)
{
    float r0 = __builtin_amdgcn_rcpf(x);
    float e = fmaf(-x, r0, 1.0f);
    float r = fmaf(e, r0, r0);
    bad = bad || !__builtin_amdgcn_classf(r, 0x108);             // not +-normal
    return r;
}
__device__ __forceinline__ float exact_rcp(float x)              // == 1.0f / x for every x (tools/rcp_exact.hip)
{
    bool bad = false;
    float r = fast_rcp_checked(x, bad);
    if (__builtin_expect(bad, 0)) r = 1.0f / x;
    return r;
}

// rt.comp:153-173 with e1 = p2-p1, e2 = p3-p1 precomputed (one rounding each, identical to the
// shader's own subtraction).  Returns hasHit; t and det are always written.
// FAST: reciprocal through fast_rcp_checked (sets `bad` when the caller must redo the test with FAST = false).
// hl: HIT_LIMIT (rt.comp:10; 0 in rt.frag:166), wave-uniform.
template <bool FAST>
__device__ __forceinline__ bool hit_triangle_t(v3 p1, v3 e1, v3 e2, v3 o, v3 d, float &t_out, float &det_out, bool &bad, float hl)
{
    v3 q = cross(d, e2);
    float det = dot(e1, q);
    float inv = FAST ? fast_rcp_checked(det, bad) : 1.0f / det;
    v3 s = o - p1;
    float u = inv * dot(s, q);
    v3 r = cross(s, e1);
    float v = inv * dot(d, r);
    float t = inv * dot(e2, r);
    t_out = t;
    det_out = det;
    // rt.comp:173: t > HIT_LIMIT && !(u < 0 || u > 1) && !(v < 0 || u + v > 1); the `det` clause there is
    // vacuous.  NaN u/v pass the negated tests as in GLSL.  IEEE minNum/maxNum ignore a NaN operand, so
    // !(min(u,v) < 0) == !(u < 0) && !(v < 0) and !(max(u,u+v) > 1) == !(u > 1) && !(u+v > 1), NaNs included.
#if PT_MINMAX
    const float lo = __builtin_fminf(u, v), hi = __builtin_fmaxf(u, u + v);
    return t > hl && !(lo < 0.0f) && !(hi > 1.0f);
#else
    return t > hl && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f);
#endif
}
__device__ __forceinline__ bool hit_triangle_ieee(v3 p1, v3 e1, v3 e2, v3 o, v3 d, float &t_out, float &det_out, float hl)
{
    bool bad = false;
    return hit_triangle_t<false>(p1, e1, e2, o, d, t_out, det_out, bad, hl);
}
// per-lane form: the fast reciprocal where it is exact, the IEEE divide for the lanes where it is not
__device__ __forceinline__ bool hit_triangle(v3 p1, v3 e1, v3 e2, v3 o, v3 d, float &t_out, float &det_out, float hl)
{
    bool bad = false;
    bool has = hit_triangle_t<true>(p1, e1, e2, o, d, t_out, det_out, bad, hl);
    if (__builtin_expect(bad, 0)) has = hit_triangle_ieee(p1, e1, e2, o, d, t_out, det_out, hl);
    return has;
}

// rt.comp:126-146; pow(length(v),2) folds to dot(v,v) and pow(r,2) to r*r in the reference's compiler.
// hl: HIT_LIMIT; xyz: the squared lengths summed as rt.frag:89-91 writes them, (x*x + y*y) + z*z, instead of dot's
// (z*z + y*y) + x*x -- both wave-uniform.
__device__ __forceinline__ float length_squared_xyz(v3 v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }
__device__ __forceinline__ bool hit_sphere(v3 c, float radius, v3 o, v3 d, float &t_out, bool &front_out, float hl, bool xyz)
{
    v3 oc = o - c;
    float a = xyz ? length_squared_xyz(d) : dot(d, d);
    float hb = dot(oc, d);
    float cc = (xyz ? length_squared_xyz(oc) : dot(oc, oc)) - radius * radius;
    float disc = hb * hb - a * cc;
    float sq = exact_sqrt(disc);
    float t0 = (-hb - sq) / a;
    float t1 = (-hb + sq) / a;
    bool front = t0 > 0.0f;
    t_out = front ? t0 : t1;
    front_out = front;
    return t0 > hl || t1 > -hl;
}

// rt.comp:235-254
__device__ __forceinline__ float fresnel_reflect_amount(float n1, float n2, v3 normal, v3 incident, float reflectivity)
{
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cosX = -dot(normal, incident);
    if (n1 > n2) {
        float n = n1 / n2;
        float sinT2 = (n * n) * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = exact_sqrt(1.0f - sinT2);
    }
    float x = 1.0f - cosX;
    float ret = r0 + (((((1.0f - r0) * x) * x) * x) * x) * x;
    return reflectivity + (1.0f - reflectivity) * ret;
}

// The same with r0 = ((n1-n2)/(n1+n2))^2 and n = n1/n2 supplied (they depend on the material and the face only).
__device__ __forceinline__ float fresnel_reflect_amount_pre(float r0, bool n1_gt_n2, float n, v3 normal, v3 incident, float reflectivity)
{
    float cosX = -dot(normal, incident);
    if (n1_gt_n2) {
        float sinT2 = (n * n) * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = exact_sqrt(1.0f - sinT2);
    }
    float x = 1.0f - cosX;
    float ret = r0 + (((((1.0f - r0) * x) * x) * x) * x) * x;
    return reflectivity + (1.0f - reflectivity) * ret;
}

// a / b for an integer-valued b in [1, 256] (`incoming / bounces`, rt.comp:322): v_rcp_f32 + one residual correction is the
// IEEE quotient whenever the result is a normal number -- every a x b = 1..256 compared by tools/sqrt_exact.hip; +0 / b
// = +0 holds trivially.  Other lanes (denormal or non-finite results, -0, b out of range) take the IEEE division.
__device__ __forceinline__ v3 div_small_int(v3 a, float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    const float x0 = a.x * r, y0 = a.y * r, z0 = a.z * r;
    v3 q = V(fmaf(fmaf(-b, x0, a.x), r, x0), fmaf(fmaf(-b, y0, a.y), r, y0), fmaf(fmaf(-b, z0, a.z), r, z0));
    const bool ok = (__builtin_amdgcn_classf(q.x, 0x108) || __float_as_uint(a.x) == 0u) && (__builtin_amdgcn_classf(q.y, 0x108) || __float_as_uint(a.y) == 0u)
                    && (__builtin_amdgcn_classf(q.z, 0x108) || __float_as_uint(a.z) == 0u) && b >= 1.0f && b <= 256.0f;
    if (__builtin_expect(!ok, 0)) q = V(a.x / b, a.y / b, a.z / b);
    return q;
}

__device__ __forceinline__ v3 reflect3(v3 i, v3 n) { float k = 2.0f * dot(n, i); return i - n * k; }
// GLSL refract as Mesa builds it: k = 1 - eta*(eta*(1 - d*d))
__device__ __forceinline__ v3 refract3(v3 i, v3 n, float eta)
{
    float d = dot(n, i);
    float k = 1.0f - eta * (eta * (1.0f - d * d));
    if (k < 0.0f) return V(0.0f, 0.0f, 0.0f);
    float c = eta * d + exact_sqrt(k);
    return i * eta - n * c;
}

}  // namespace pt
