// Math.h — the few vector/matrix operations the host classes need, float32, written so that
// Camera::UpdateView reproduces the reference's matrices bit for bit (it used GLM 0.9.8.5:
// dependencies/glm/gtc/matrix_transform.inl:521-546 lookAtRH, detail/func_matrix.inl:297-356 inverse).
// Own code; no GLM here.  Compile with -ffp-contract=off.
#pragma once
#include <cmath>

namespace ptm {

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float a) : x(a), y(a), z(a) {}
    vec3(double a, double b, double c) : x((float)a), y((float)b), z((float)c) {}
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    vec4(float a) : x(a), y(a), z(a), w(a) {}
    vec4(double a, double b, double c, double d) : x((float)a), y((float)b), z((float)c), w((float)d) {}
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct ivec4 { int x = 0, y = 0, z = 0, w = 0; };

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }
inline vec3 &operator-=(vec3 &a, vec3 b) { a = a - b; return a; }
inline vec4 operator*(vec4 a, vec4 b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(vec4 a, vec4 b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 min(vec4 a, vec4 b) { return vec4(b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z, b.w < a.w ? b.w : a.w); }
inline vec4 max(vec4 a, vec4 b) { return vec4(a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z, a.w < b.w ? b.w : a.w); }

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline float dot(vec3 a, vec3 b) { float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return (tx + ty) + tz; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

struct mat4 {
    vec4 c[4];                       // columns, like GLM and like glUniformMatrix4fv(..., GL_FALSE, ...)
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
    const float *data() const { return &c[0].x; }
};

inline mat4 lookAt(vec3 eye, vec3 center, vec3 up)      // right-handed
{
    const vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
    mat4 m;
    m[0] = vec4(s.x, u.x, -f.x, 0); m[1] = vec4(s.y, u.y, -f.y, 0); m[2] = vec4(s.z, u.z, -f.z, 0);
    m[3] = vec4(-dot(s, eye), -dot(u, eye), dot(f, eye), 1);
    return m;
}

inline mat4 inverse(const mat4 &m)                       // general cofactor inverse, GLM's operation order
{
    auto c2 = [](float a, float b, float c, float d) { return a * b - c * d; };
    const float k00 = c2(m[2][2], m[3][3], m[3][2], m[2][3]), k02 = c2(m[1][2], m[3][3], m[3][2], m[1][3]), k03 = c2(m[1][2], m[2][3], m[2][2], m[1][3]);
    const float k04 = c2(m[2][1], m[3][3], m[3][1], m[2][3]), k06 = c2(m[1][1], m[3][3], m[3][1], m[1][3]), k07 = c2(m[1][1], m[2][3], m[2][1], m[1][3]);
    const float k08 = c2(m[2][1], m[3][2], m[3][1], m[2][2]), k10 = c2(m[1][1], m[3][2], m[3][1], m[1][2]), k11 = c2(m[1][1], m[2][2], m[2][1], m[1][2]);
    const float k12 = c2(m[2][0], m[3][3], m[3][0], m[2][3]), k14 = c2(m[1][0], m[3][3], m[3][0], m[1][3]), k15 = c2(m[1][0], m[2][3], m[2][0], m[1][3]);
    const float k16 = c2(m[2][0], m[3][2], m[3][0], m[2][2]), k18 = c2(m[1][0], m[3][2], m[3][0], m[1][2]), k19 = c2(m[1][0], m[2][2], m[2][0], m[1][2]);
    const float k20 = c2(m[2][0], m[3][1], m[3][0], m[2][1]), k22 = c2(m[1][0], m[3][1], m[3][0], m[1][1]), k23 = c2(m[1][0], m[2][1], m[2][0], m[1][1]);
    const vec4 f0(k00, k00, k02, k03), f1(k04, k04, k06, k07), f2(k08, k08, k10, k11), f3(k12, k12, k14, k15), f4(k16, k16, k18, k19), f5(k20, k20, k22, k23);
    const vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]), v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    const vec4 i0 = (v1 * f0 - v2 * f1) + v3 * f2, i1 = (v0 * f0 - v2 * f3) + v3 * f4, i2 = (v0 * f1 - v1 * f3) + v3 * f5, i3 = (v0 * f2 - v1 * f4) + v2 * f5;
    const vec4 sa(1, -1, 1, -1), sb(-1, 1, -1, 1);
    mat4 inv;
    inv[0] = i0 * sa; inv[1] = i1 * sb; inv[2] = i2 * sa; inv[3] = i3 * sb;
    const vec4 d0 = m[0] * vec4(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    const float det = (d0.x + d0.y) + (d0.z + d0.w);
    const float r = 1.0f / det;
    for (int k = 0; k < 4; k++) inv[k] = inv[k] * r;
    return inv;
}

}  // namespace ptm
