// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; nothing under
// vk-renderer_amd/ may include, link or call this.  Parity: UNPINNED by the
// reference (it ships no tests / goldens and cannot be built here).
//
// glsl.hpp — the GLSL 4.60 value semantics the restated shaders rely on, frozen to
// one IEEE-754 binary32 operation sequence so that a GPU implementation following
// the same sequence is bit-identical for everything that feeds a branch.
//
// NUMERIC CONTRACT, version VKR_CONTRACT (default 2; build with CONTRACT=1 for the first contract, to bisect).
//   GLSL without `precise` lets every Vulkan driver contract a*b+c into one fused multiply-add, so a restatement that
//   fuses is as faithful to the reference as one that does not — but oracle and kernels must fuse the SAME expressions.
//   Contract 1 fused nothing.  Contract 2 fuses exactly the accumulation steps of the helpers below, written with
//   cfma(a,b,c) (one IEEE fma under contract 2, round(a*b)+c under contract 1, same association order either way):
//   dot, mix, mat4*vec4, cross, reflect, madd(a,s,b) = a + s*b, the texel coordinate uv*size - 0.5 of the sampler,
//   2x-1 / 0.5x+0.5 range maps, d*(f-n)-f of linearize_depth2.  Everything else keeps one rounding per operation.
//   The compilers never contract on their own: -ffp-contract=off on both sides.
//
// Frozen choices (GLSL leaves them implementation-defined):
//   dot(a,b)       = cfma(a.z,b.z, cfma(a.y,b.y, a.x*b.x)) [then a.w*b.w]
//   normalize(v)   = v * (1.0f / sqrtf(dot(v,v)))
//   length(v)      = sqrtf(dot(v,v))
//   mix(a,b,t)     = cfma(b, t, a*(1-t))      (GLSL spec formula a*(1-t) + b*t)
//   min/max        = IEEE minNum/maxNum (fminf/fmaxf; what v_min_f32/v_max_f32 do)
//   clamp(x,lo,hi) = min(max(x,lo),hi)
//   reflect(I,N)   = I - (2*dot(N,I))*N
//   int(float)     = truncation, NaN -> 0, saturating at +-2^30
//   mat4*vec4      = column-major, sum left to right
// Build with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef VKR_CONTRACT
#define VKR_CONTRACT 2
#endif

namespace glsl {

// the one place where the two numeric contracts differ
inline float cfma(float a, float b, float c) {
#if VKR_CONTRACT >= 2
  return __builtin_fmaf(a, b, c);
#else
  return a * b + c;
#endif
}

struct vec2 { float x, y; vec2() : x(0), y(0) {} vec2(float a, float b) : x(a), y(b) {} explicit vec2(float a) : x(a), y(a) {} };
struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit vec3(float a) : x(a), y(a), z(a) {}
  vec3(vec2 v, float c) : x(v.x), y(v.y), z(c) {}
  vec2 xy() const { return vec2(x, y); }
};
struct vec4 {
  float x, y, z, w;
  vec4() : x(0), y(0), z(0), w(0) {}
  vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
  vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
  vec2 xy() const { return vec2(x, y); }
  vec3 xyz() const { return vec3(x, y, z); }
};
struct ivec2 { int x, y; ivec2() : x(0), y(0) {} ivec2(int a, int b) : x(a), y(b) {} };

#define GLSL_BINOP(V, op)                                                                   \
  inline V operator op(V a, V b);                                                           \
  inline V operator op(V a, float s);                                                       \
  inline V operator op(float s, V a);
GLSL_BINOP(vec2, +) GLSL_BINOP(vec2, -) GLSL_BINOP(vec2, *) GLSL_BINOP(vec2, /)
GLSL_BINOP(vec3, +) GLSL_BINOP(vec3, -) GLSL_BINOP(vec3, *) GLSL_BINOP(vec3, /)
GLSL_BINOP(vec4, +) GLSL_BINOP(vec4, -) GLSL_BINOP(vec4, *) GLSL_BINOP(vec4, /)
#undef GLSL_BINOP

#define GLSL_DEF2(op)                                                                       \
  inline vec2 operator op(vec2 a, vec2 b) { return vec2(a.x op b.x, a.y op b.y); }          \
  inline vec2 operator op(vec2 a, float s) { return vec2(a.x op s, a.y op s); }             \
  inline vec2 operator op(float s, vec2 a) { return vec2(s op a.x, s op a.y); }
#define GLSL_DEF3(op)                                                                       \
  inline vec3 operator op(vec3 a, vec3 b) { return vec3(a.x op b.x, a.y op b.y, a.z op b.z); } \
  inline vec3 operator op(vec3 a, float s) { return vec3(a.x op s, a.y op s, a.z op s); }   \
  inline vec3 operator op(float s, vec3 a) { return vec3(s op a.x, s op a.y, s op a.z); }
#define GLSL_DEF4(op)                                                                       \
  inline vec4 operator op(vec4 a, vec4 b) { return vec4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); } \
  inline vec4 operator op(vec4 a, float s) { return vec4(a.x op s, a.y op s, a.z op s, a.w op s); } \
  inline vec4 operator op(float s, vec4 a) { return vec4(s op a.x, s op a.y, s op a.z, s op a.w); }
GLSL_DEF2(+) GLSL_DEF2(-) GLSL_DEF2(*) GLSL_DEF2(/)
GLSL_DEF3(+) GLSL_DEF3(-) GLSL_DEF3(*) GLSL_DEF3(/)
GLSL_DEF4(+) GLSL_DEF4(-) GLSL_DEF4(*) GLSL_DEF4(/)
#undef GLSL_DEF2
#undef GLSL_DEF3
#undef GLSL_DEF4

inline vec2 operator-(vec2 a) { return vec2(-a.x, -a.y); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec2& operator+=(vec2& a, vec2 b) { a = a + b; return a; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }
inline vec3& operator/=(vec3& a, vec3 b) { a = a / b; return a; }
inline vec2& operator*=(vec2& a, float s) { a = a * s; return a; }
inline ivec2 operator+(ivec2 a, ivec2 b) { return ivec2(a.x + b.x, a.y + b.y); }
inline ivec2 operator*(int s, ivec2 a) { return ivec2(s * a.x, s * a.y); }

inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline int   min(int a, int b) { return a < b ? a : b; }
inline int   max(int a, int b) { return a > b ? a : b; }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline int   clamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
inline float abs(float a) { return fabsf(a); }
inline vec2  abs(vec2 a) { return vec2(fabsf(a.x), fabsf(a.y)); }
inline vec3  min(vec3 a, vec3 b) { return vec3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
inline vec3  max(vec3 a, vec3 b) { return vec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline vec3  clamp(vec3 v, vec3 lo, vec3 hi) { return min(max(v, lo), hi); }
inline float floor(float a) { return floorf(a); }
inline vec2  floor(vec2 a) { return vec2(floorf(a.x), floorf(a.y)); }
inline float fract(float a) { return a - floorf(a); }
inline float mix(float a, float b, float t) { return cfma(b, t, a * (1.0f - t)); }
inline vec2  mix(vec2 a, vec2 b, float t) { return vec2(mix(a.x, b.x, t), mix(a.y, b.y, t)); }
inline vec3  mix(vec3 a, vec3 b, float t) { return vec3(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)); }
inline vec4  mix(vec4 a, vec4 b, float t) { return vec4(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t), mix(a.w, b.w, t)); }

inline float dot(vec2 a, vec2 b) { return cfma(a.y, b.y, a.x * b.x); }
inline float dot(vec3 a, vec3 b) { return cfma(a.z, b.z, cfma(a.y, b.y, a.x * b.x)); }
inline float dot(vec4 a, vec4 b) { return cfma(a.w, b.w, cfma(a.z, b.z, cfma(a.y, b.y, a.x * b.x))); }
inline float length(vec2 a) { return sqrtf(dot(a, a)); }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline vec2  normalize(vec2 a) { return a * (1.0f / sqrtf(dot(a, a))); }
inline vec3  normalize(vec3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
inline vec3  cross(vec3 a, vec3 b) {
  return vec3(cfma(a.y, b.z, -(a.z * b.y)), cfma(a.z, b.x, -(a.x * b.z)), cfma(a.x, b.y, -(a.y * b.x)));
}
// a + s * b, the "advance along a direction" shape (ray positions, sample positions, projections off a normal)
inline vec2 madd(vec2 a, float s, vec2 b) { return vec2(cfma(s, b.x, a.x), cfma(s, b.y, a.y)); }
inline vec3 madd(vec3 a, float s, vec3 b) { return vec3(cfma(s, b.x, a.x), cfma(s, b.y, a.y), cfma(s, b.z, a.z)); }
inline vec3 reflect(vec3 I, vec3 N) { return madd(I, -(2.0f * dot(N, I)), N); }
inline bool isnan(float a) { return a != a; }

// float -> int conversion: truncation toward zero, NaN -> 0, saturating at +-2^30 (so that
// a following +1 cannot overflow; any such coordinate is far outside every image).
inline int f2i(float f) {
  if (f != f) return 0;
  return (int)fminf(fmaxf(f, -1073741824.0f), 1073741824.0f);
}
inline uint32_t f2u(float f) {
  if (f != f) return 0u;
  return (uint32_t)fminf(fmaxf(f, 0.0f), 1073741824.0f);
}
inline ivec2 to_ivec2(vec2 v) { return ivec2(f2i(v.x), f2i(v.y)); }

struct mat4 {
  float m[16];  // column-major: m[c*4 + r]
  float at(int r, int c) const { return m[c * 4 + r]; }
};
inline vec4 operator*(const mat4& M, vec4 v) {
  vec4 r;
  r.x = cfma(M.m[12], v.w, cfma(M.m[8], v.z, cfma(M.m[4], v.y, M.m[0] * v.x)));
  r.y = cfma(M.m[13], v.w, cfma(M.m[9], v.z, cfma(M.m[5], v.y, M.m[1] * v.x)));
  r.z = cfma(M.m[14], v.w, cfma(M.m[10], v.z, cfma(M.m[6], v.y, M.m[2] * v.x)));
  r.w = cfma(M.m[15], v.w, cfma(M.m[11], v.z, cfma(M.m[7], v.y, M.m[3] * v.x)));
  return r;
}

static const float PI = 3.1415926535897932384626433832795f;

}  // namespace glsl
