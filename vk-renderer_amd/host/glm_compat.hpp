// glm_compat.hpp — the slice of glm the hot-path pass structs use (mat4 / vec4 storage,
// inverse, transpose, perspective, lookAt).  glm is a system dependency of the reference
// that is not vendored; when <glm/glm.hpp> is available it is used, with the
// reference's own configuration (scene/camera.hpp:4-5).  Column-major, memory-compatible.
#ifndef VKR_GLM_COMPAT_HPP_INCLUDED
#define VKR_GLM_COMPAT_HPP_INCLUDED
#if __has_include(<glm/glm.hpp>) && !defined(VKR_FORCE_GLM_COMPAT)
#define GLM_FORCE_RADIANS
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#else
#include <cmath>
namespace glm {
struct vec2 { float x = 0, y = 0; vec2() {} vec2(float a, float b) : x(a), y(b) {} };
struct vec3 { float x = 0, y = 0, z = 0; vec3() {} vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; } };
struct vec4 { float x = 0, y = 0, z = 0, w = 0; vec4() {} vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
  float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; } };
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline vec3 normalize(vec3 a) { float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline float length(vec4 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w); }
inline float floor(float a) { return std::floor(a); }
inline vec4& operator/=(vec4& a, float s) { a.x /= s; a.y /= s; a.z /= s; a.w /= s; return a; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
struct mat4 {
  vec4 c[4];  // columns
  mat4() : mat4(1.f) {}
  explicit mat4(float d) { c[0].x = d; c[1].y = d; c[2].z = d; c[3].w = d; }
  vec4& operator[](int i) { return c[i]; }
  const vec4& operator[](int i) const { return c[i]; }
};
inline mat4 operator*(const mat4& a, const mat4& b) {
  mat4 r(0.f);
  for (int col = 0; col < 4; col++)
    for (int row = 0; row < 4; row++) {
      double s = 0.0;  // accumulate in double, round once
      for (int k = 0; k < 4; k++) s += (double)a[k][row] * (double)b[col][k];
      r[col][row] = (float)s;
    }
  return r;
}
inline vec4 operator*(const mat4& a, const vec4& v) {
  vec4 r;
  for (int row = 0; row < 4; row++)
    r[row] = (float)((double)a[0][row] * v.x + (double)a[1][row] * v.y + (double)a[2][row] * v.z + (double)a[3][row] * v.w);
  return r;
}
inline mat4 transpose(const mat4& m) { mat4 r(0.f); for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r[i][j] = m[j][i]; return r; }
// general 4x4 inverse by cofactors, evaluated in double
inline mat4 inverse(const mat4& m) {
  double a[16], inv[16];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) a[i * 4 + j] = m[i][j];
  inv[0] = a[5]*a[10]*a[15] - a[5]*a[11]*a[14] - a[9]*a[6]*a[15] + a[9]*a[7]*a[14] + a[13]*a[6]*a[11] - a[13]*a[7]*a[10];
  inv[4] = -a[4]*a[10]*a[15] + a[4]*a[11]*a[14] + a[8]*a[6]*a[15] - a[8]*a[7]*a[14] - a[12]*a[6]*a[11] + a[12]*a[7]*a[10];
  inv[8] = a[4]*a[9]*a[15] - a[4]*a[11]*a[13] - a[8]*a[5]*a[15] + a[8]*a[7]*a[13] + a[12]*a[5]*a[11] - a[12]*a[7]*a[9];
  inv[12] = -a[4]*a[9]*a[14] + a[4]*a[10]*a[13] + a[8]*a[5]*a[14] - a[8]*a[6]*a[13] - a[12]*a[5]*a[10] + a[12]*a[6]*a[9];
  inv[1] = -a[1]*a[10]*a[15] + a[1]*a[11]*a[14] + a[9]*a[2]*a[15] - a[9]*a[3]*a[14] - a[13]*a[2]*a[11] + a[13]*a[3]*a[10];
  inv[5] = a[0]*a[10]*a[15] - a[0]*a[11]*a[14] - a[8]*a[2]*a[15] + a[8]*a[3]*a[14] + a[12]*a[2]*a[11] - a[12]*a[3]*a[10];
  inv[9] = -a[0]*a[9]*a[15] + a[0]*a[11]*a[13] + a[8]*a[1]*a[15] - a[8]*a[3]*a[13] - a[12]*a[1]*a[11] + a[12]*a[3]*a[9];
  inv[13] = a[0]*a[9]*a[14] - a[0]*a[10]*a[13] - a[8]*a[1]*a[14] + a[8]*a[2]*a[13] + a[12]*a[1]*a[10] - a[12]*a[2]*a[9];
  inv[2] = a[1]*a[6]*a[15] - a[1]*a[7]*a[14] - a[5]*a[2]*a[15] + a[5]*a[3]*a[14] + a[13]*a[2]*a[7] - a[13]*a[3]*a[6];
  inv[6] = -a[0]*a[6]*a[15] + a[0]*a[7]*a[14] + a[4]*a[2]*a[15] - a[4]*a[3]*a[14] - a[12]*a[2]*a[7] + a[12]*a[3]*a[6];
  inv[10] = a[0]*a[5]*a[15] - a[0]*a[7]*a[13] - a[4]*a[1]*a[15] + a[4]*a[3]*a[13] + a[12]*a[1]*a[7] - a[12]*a[3]*a[5];
  inv[14] = -a[0]*a[5]*a[14] + a[0]*a[6]*a[13] + a[4]*a[1]*a[14] - a[4]*a[2]*a[13] - a[12]*a[1]*a[6] + a[12]*a[2]*a[5];
  inv[3] = -a[1]*a[6]*a[11] + a[1]*a[7]*a[10] + a[5]*a[2]*a[11] - a[5]*a[3]*a[10] - a[9]*a[2]*a[7] + a[9]*a[3]*a[6];
  inv[7] = a[0]*a[6]*a[11] - a[0]*a[7]*a[10] - a[4]*a[2]*a[11] + a[4]*a[3]*a[10] + a[8]*a[2]*a[7] - a[8]*a[3]*a[6];
  inv[11] = -a[0]*a[5]*a[11] + a[0]*a[7]*a[9] + a[4]*a[1]*a[11] - a[4]*a[3]*a[9] - a[8]*a[1]*a[7] + a[8]*a[3]*a[5];
  inv[15] = a[0]*a[5]*a[10] - a[0]*a[6]*a[9] - a[4]*a[1]*a[10] + a[4]*a[2]*a[9] + a[8]*a[1]*a[6] - a[8]*a[2]*a[5];
  double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
  mat4 r(0.f);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r[i][j] = (float)(inv[i * 4 + j] / det);
  return r;
}
// RH, depth 0..1 (GLM_FORCE_DEPTH_ZERO_TO_ONE, scene/camera.hpp:4-5)
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar) {
  const float t = std::tan(fovy / 2.f);
  mat4 r(0.f);
  r[0][0] = 1.f / (aspect * t);
  r[1][1] = 1.f / t;
  r[2][2] = zFar / (zNear - zFar);
  r[2][3] = -1.f;
  r[3][2] = -(zFar * zNear) / (zFar - zNear);
  return r;
}
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
  const vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
  mat4 r(1.f);
  r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
  r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
  r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
  r[3][0] = -dot(s, eye); r[3][1] = -dot(u, eye); r[3][2] = dot(f, eye);
  return r;
}
}  // namespace glm
#endif
#endif
