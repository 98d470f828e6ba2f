// vkr_device.hpp — device-side vocabulary of the HIP hot path (gfx950).
//
// * f2/f3/f4 value types with one frozen IEEE-754 binary32 operation order so hit/no-hit
//   and horizon-break decisions are reproducible bit-for-bit.  NUMERIC CONTRACT version
//   VKR_CONTRACT (default 2; `make CONTRACT=1` builds the first one, to bisect): GLSL without
//   `precise` lets a Vulkan driver fuse a*b+c, so fusing is as faithful to the reference as not
//   fusing, provided oracle and kernels fuse the SAME expressions.  Contract 2 fuses exactly the
//   accumulation steps of dot, mix, mat4*vec4, cross, reflect, madd(a,s,b) = a + s*b, the
//   sampler's texel coordinate uv*size - 0.5, the 2x-1 / 0.5x+0.5 range maps and d*(f-n)-f of
//   linearize_depth2, all written with cfma() below; contract 1 fused nothing.  The compiler
//   never contracts on its own (-ffp-contract=off);
// * Tex: one mip of a pitch-linear image window in HBM; typed texel loaders for the
//   reference's storage formats (scene_renderer.cpp:13-43, gtao.cpp:26-47,
//   advanced_ssr.cpp:62-92, taa.cpp:6);
// * bilinear / texelFetch with gpu::DEFAULT_SAMPLER semantics
//   (src/gpu/samplers.hpp:36-55): linear, clamp-to-edge, LOD 0 of the bound view;
//   out-of-frame texelFetch returns 0.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#ifndef VKR_CONTRACT
#define VKR_CONTRACT 2
#endif

namespace vkr {

#define VKR_DEV __device__ __forceinline__

// the one place where the two numeric contracts differ: a fused multiply-add (contract 2) or round(a*b) + c (contract 1);
// every helper keeps the same association order under both
VKR_DEV float cfma(float a, float b, float c) {
#if VKR_CONTRACT >= 2
  return __builtin_fmaf(a, b, c);
#else
  return a * b + c;
#endif
}

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct i2 { int x, y; };

VKR_DEV f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
VKR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VKR_DEV f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
VKR_DEV f3 xyz(f4 v) { return mk3(v.x, v.y, v.z); }
VKR_DEV f2 xy(f3 v) { return mk2(v.x, v.y); }
VKR_DEV f2 xy(f4 v) { return mk2(v.x, v.y); }

VKR_DEV f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
VKR_DEV f2 operator-(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
VKR_DEV f2 operator*(f2 a, f2 b) { return mk2(a.x * b.x, a.y * b.y); }
VKR_DEV f2 operator/(f2 a, f2 b) { return mk2(a.x / b.x, a.y / b.y); }
VKR_DEV f2 operator*(f2 a, float s) { return mk2(a.x * s, a.y * s); }
VKR_DEV f2 operator*(float s, f2 a) { return mk2(s * a.x, s * a.y); }
VKR_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VKR_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VKR_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VKR_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
VKR_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VKR_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
VKR_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
VKR_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VKR_DEV f4 operator/(f4 a, float s) { return mk4(a.x / s, a.y / s, a.z / s, a.w / s); }

// IEEE minNum / maxNum (v_min_f32 / v_max_f32)
VKR_DEV float vmin(float a, float b) { return fminf(a, b); }
VKR_DEV float vmax(float a, float b) { return fmaxf(a, b); }
VKR_DEV float vclamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
VKR_DEV int iclamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
VKR_DEV float mixf(float a, float b, float t) { return cfma(b, t, a * (1.0f - t)); }
VKR_DEV f2 mix2(f2 a, f2 b, float t) { return mk2(mixf(a.x, b.x, t), mixf(a.y, b.y, t)); }
VKR_DEV f3 mix3(f3 a, f3 b, float t) { return mk3(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }
VKR_DEV f4 mix4(f4 a, f4 b, float t) { return mk4(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t), mixf(a.w, b.w, t)); }
VKR_DEV f3 min3(f3 a, f3 b) { return mk3(vmin(a.x, b.x), vmin(a.y, b.y), vmin(a.z, b.z)); }
VKR_DEV f3 max3(f3 a, f3 b) { return mk3(vmax(a.x, b.x), vmax(a.y, b.y), vmax(a.z, b.z)); }
VKR_DEV float dot(f2 a, f2 b) { return cfma(a.y, b.y, a.x * b.x); }
VKR_DEV float dot(f3 a, f3 b) { return cfma(a.z, b.z, cfma(a.y, b.y, a.x * b.x)); }
// a + s * b: ray positions, sample positions, projections off a normal
VKR_DEV f2 madd(f2 a, float s, f2 b) { return mk2(cfma(s, b.x, a.x), cfma(s, b.y, a.y)); }
VKR_DEV f3 madd(f3 a, float s, f3 b) { return mk3(cfma(s, b.x, a.x), cfma(s, b.y, a.y), cfma(s, b.z, a.z)); }
// a / b for finite operands in the normal range (no scaling, no special cases): the refinement
// sequence the compiler emits for IEEE division minus v_div_scale / v_div_fixup — same correctly
// rounded result (vkr_selftest_division checks it on the GPU), ~24 instead of ~43 cycles.
VKR_DEV float div_normal(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = a * r;
  const float e2 = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(e2, r, q);
  const float e3 = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(e3, r, q);
}
// sqrtf(x), correctly rounded, without the compiler's input scaling (v_sqrt_f32 does not take denormals) and class checks:
// v_sqrt_f32 is within 1 ulp, the two neighbours are tried with exact residuals (the compiler's own refinement).  Valid for
// x >= 2^-96 (the results of every dot product of this path); anything else — 0, tiny, negative, NaN — takes the library
// sequence.  vkr_selftest_sqrt compares it with sqrtf on EVERY float of the fast range.
VKR_DEV float sqrt_ieee(float x) {
  if (__builtin_expect(!(x >= 0x1p-96f), 0)) return sqrtf(x);
  float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float e_dn = __builtin_fmaf(-s_dn, s, x), e_up = __builtin_fmaf(-s_up, s, x);
  s = e_dn <= 0.0f ? s_dn : s;
  s = e_up > 0.0f ? s_up : s;
  return s;
}
// 1.0f / s for the s = sqrt_ieee(x) of a vector length (2^-48 <= s <= 2^64: operands and quotient are normal)
VKR_DEV float rcp_ieee_normal(float s) { return div_normal(1.0f, s); }
VKR_DEV float length(f2 a) { return sqrt_ieee(dot(a, a)); }
VKR_DEV float length(f3 a) { return sqrt_ieee(dot(a, a)); }
// v * (1 / sqrt(v . v)): the two correctly rounded operations of the contract on their cheap exact forms; a zero or
// denormal-length vector (never on this path) keeps the IEEE special cases
VKR_DEV f3 normalize(f3 a) {
  const float d = dot(a, a);
  if (__builtin_expect(!(d >= 0x1p-96f && d <= 0x1p96f), 0)) return a * (1.0f / sqrtf(d));
  return a * rcp_ieee_normal(sqrt_ieee(d));
}
// Hardware reciprocal / rsqrt / sqrt (1 ulp): ONLY for values that never feed a comparison — weights,
// shading terms, horizon cosines that are max()-reduced — where 1e-7 relative noise is irrelevant.
VKR_DEV float fast_rcp(float a) { return __builtin_amdgcn_rcpf(a); }
VKR_DEV float fast_rsq(float a) { return __builtin_amdgcn_rsqf(a); }
VKR_DEV float fast_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
VKR_DEV f3 normalize_fast(f3 a) { return a * fast_rsq(dot(a, a)); }
VKR_DEV f3 cross(f3 a, f3 b) { return mk3(cfma(a.y, b.z, -(a.z * b.y)), cfma(a.z, b.x, -(a.x * b.z)), cfma(a.x, b.y, -(a.y * b.x))); }
VKR_DEV f3 reflect(f3 I, f3 N) { return madd(I, -(2.0f * dot(N, I)), N); }
VKR_DEV bool is_nan(float a) { return a != a; }
VKR_DEV float fractf(float a) { return a - floorf(a); }
// float -> int: truncate, NaN -> 0, saturate at +-2^30 (a following +1 cannot overflow).
// v_cvt_i32_f32 itself truncates, saturates and maps NaN to 0; the med3 narrows the saturation.
VKR_DEV int f2i(float f) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return min(max(r, -1073741824), 1073741824);
}
// float -> int for an index that is only range-checked: v_cvt_i32_f32 as it is (truncate, NaN -> 0, saturate at INT_MIN /
// INT_MAX, both of which fail an unsigned `< extent` test)
VKR_DEV int f2i_index(float f) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
VKR_DEV uint32_t f2u(float f) {
  uint32_t r;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
  return min(r, 1073741824u);
}

// XCD-aware tile order (after cdna_hip_programming.md 5.5 T1).  The dispatcher deals consecutive
// workgroup ids round-robin over the 8 XCDs, each with a private 4 MiB L2, so with the plain
// blockIdx -> tile map every neighbour of a tile runs on another XCD and re-fetches the shared apron /
// bilinear footprint over the fabric (measured: 2.2-3.6x the algorithmic bytes).  Here the grid is cut
// into chunks of CW x CH tiles; the T = CW*CH ids of one residue class mod 8 (= one XCD) inside a run
// of 8T ids cover one chunk, consecutive chunks go to consecutive XCDs.  Small chunks (not one band
// per XCD) keep the per-XCD load balanced: tile cost varies 10x across the frame (sky vs. rough
// surfaces).  Ids past the last full group of 8 chunks, and tiles outside the chunked region, map in
// plain order.  A bijection for every grid size; pure speed — any placement gives the same result.
template <int CW, int CH> VKR_DEV i2 xcd_tile(unsigned id, unsigned GW, unsigned GH) {
  const unsigned T = CW * CH;
  const unsigned ncx = GW / CW, ncy = GH / CH;
  const unsigned R = ncx * ncy * T;             // tiles inside whole chunks
  const unsigned R8 = (ncx * ncy / 8u) * 8u * T;  // ... inside whole groups of 8 chunks
  unsigned chunk, n;
  i2 b;
  if (id < R8) {
    const unsigned g = id / (8u * T), w = id % (8u * T);
    chunk = g * 8u + (w & 7u);
    n = w >> 3;
  } else if (id < R) {
    chunk = id / T;
    n = id % T;
  } else {  // leftover tiles: the right strip (rows of whole chunks), then the bottom strip
    const unsigned p = id - R, sw = GW - ncx * CW, right = sw * (ncy * CH);
    if (p < right) { b.x = (int)(ncx * CW + p % sw); b.y = (int)(p / sw); }
    else { b.x = (int)((p - right) % GW); b.y = (int)(ncy * CH + (p - right) / GW); }
    return b;
  }
  b.x = (int)((chunk % ncx) * CW + n % CW);
  b.y = (int)((chunk / ncx) * CH + n / CW);
  return b;
}
template <int CW, int CH> VKR_DEV i2 xcd_block() { return xcd_tile<CW, CH>(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y); }

#define VKR_PI 3.1415926535897932384626433832795f

struct Mat4 { float m[16]; };  // column-major
VKR_DEV f4 mul(const Mat4& M, f4 v) {
  f4 r;
  r.x = cfma(M.m[12], v.w, cfma(M.m[8], v.z, cfma(M.m[4], v.y, M.m[0] * v.x)));
  r.y = cfma(M.m[13], v.w, cfma(M.m[9], v.z, cfma(M.m[5], v.y, M.m[1] * v.x)));
  r.z = cfma(M.m[14], v.w, cfma(M.m[10], v.z, cfma(M.m[6], v.y, M.m[2] * v.x)));
  r.w = cfma(M.m[15], v.w, cfma(M.m[11], v.z, cfma(M.m[7], v.y, M.m[3] * v.x)));
  return r;
}

// ---- storage codecs -------------------------------------------------------------------
#include "srgb_tables.inc"
VKR_DEV float srgb8_to_float(uint32_t c) { return __uint_as_float(k_srgb_decode_bits[c]); }
VKR_DEV uint32_t float_to_srgb8(float x) {
  if (x != x) return 0u;
  int lo = 0, hi = 255;  // largest i with thresh[i] <= x
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (__uint_as_float(k_srgb_thresh_bits[mid]) <= x) lo = mid; else hi = mid - 1;
  }
  return (uint32_t)lo;
}
// UNORM -> float is defined as the correctly rounded quotient k / (2^b - 1).  With x = k * 2^-b
// (exact), fmaf(x, C, x) for the constants below equals that quotient for EVERY code of each width
// (exhaustively checked in tests/test_codecs.py): three full-rate fp32 ops instead of an IEEE
// division sequence (~43 cycles per wave on gfx950) or three quarter-rate f64 ops.
VKR_DEV float d24_to_float(uint32_t t) {
  const float x = (float)(t & 0xFFFFFFu) * 0x1p-24f;
  return __builtin_fmaf(x, 0x1.000002p-24f, x);  // nextafter(2^-24)
}
VKR_DEV float unorm16_to_float(uint32_t v) {
  const float x = (float)v * 0x1p-16f;
  return __builtin_fmaf(x, 0x1.0001p-16f, x);    // 2^-16 + 2^-32
}
VKR_DEV float unorm8_to_float(uint32_t v) {
  const float x = (float)v * 0x1p-8f;
  return __builtin_fmaf(x, 0x1.010102p-8f, x);   // float(1/255)
}
VKR_DEV uint32_t float_to_unorm16(float f) { return (uint32_t)rintf(vclamp(f, 0.0f, 1.0f) * 65535.0f); }
VKR_DEV uint32_t float_to_unorm8(float f) { return (uint32_t)rintf(vclamp(f, 0.0f, 1.0f) * 255.0f); }
VKR_DEV float half_bits_to_float(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }
VKR_DEV uint32_t float_to_half_bits(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }

// ---- image windows ----------------------------------------------------------------------
// One mip level of an image window.  (w,h): extent held in memory; (fw,fh): extent of the
// whole frame at this mip; (ox,oy): window origin in the frame.
struct Tex {
  const uint8_t* p;
  int pitch;
  int w, h;
  int fw, fh;
  int ox, oy;
};
// Byte offset of a texel inside its window: 32-bit (make_tex rejects windows of 4 GiB and more), so that a load from a
// kernel-argument image is `global_load v, v_offset, s[base]` — one v_mad_u32_u24-class instruction per address instead of
// a 64-bit multiply-add pair.
VKR_DEV uint32_t toff(const Tex& t, int lx, int ly, int bpp) { return __umul24((uint32_t)ly, (uint32_t)t.pitch) + (uint32_t)lx * (uint32_t)bpp; }
struct Pyramid {
  Tex mip[16];
  int count;
};

struct FmtD24 { typedef float T; static VKR_DEV T zero() { return 0.0f; }
  static VKR_DEV T decode(uint32_t v) { return d24_to_float(v); }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { return decode(*(const uint32_t*)(t.p + toff(t, lx, ly, 4))); }
  static VKR_DEV T lerp(T a, T b, float f) { return mixf(a, b, f); } };
struct FmtR32F { typedef float T; static VKR_DEV T zero() { return 0.0f; }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { return *(const float*)(t.p + toff(t, lx, ly, 4)); }
  static VKR_DEV T lerp(T a, T b, float f) { return mixf(a, b, f); } };
struct FmtR16F { typedef float T; static VKR_DEV T zero() { return 0.0f; }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { return half_bits_to_float(*(const uint16_t*)(t.p + toff(t, lx, ly, 2))); }
  static VKR_DEV T lerp(T a, T b, float f) { return mixf(a, b, f); } };
struct FmtRG16U { typedef f2 T; static VKR_DEV T zero() { return mk2(0, 0); }
  static VKR_DEV T decode(uint32_t v) { return mk2(unorm16_to_float(v & 0xFFFFu), unorm16_to_float(v >> 16)); }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { return decode(*(const uint32_t*)(t.p + toff(t, lx, ly, 4))); }
  static VKR_DEV T lerp(T a, T b, float f) { return mix2(a, b, f); } };
struct FmtRG16F { typedef f2 T; static VKR_DEV T zero() { return mk2(0, 0); }
  static VKR_DEV T decode(uint32_t v) { return mk2(half_bits_to_float(v & 0xFFFFu), half_bits_to_float(v >> 16)); }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { return decode(*(const uint32_t*)(t.p + toff(t, lx, ly, 4))); }
  static VKR_DEV T lerp(T a, T b, float f) { return mix2(a, b, f); } };
// rgb of an RGBA8_SRGB texel (alpha is never consumed on this path)
struct FmtSRGB8 { typedef f3 T; static VKR_DEV T zero() { return mk3(0, 0, 0); }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { uint32_t v = *(const uint32_t*)(t.p + toff(t, lx, ly, 4));
    return mk3(srgb8_to_float(v & 0xFFu), srgb8_to_float((v >> 8) & 0xFFu), srgb8_to_float((v >> 16) & 0xFFu)); }
  static VKR_DEV T lerp(T a, T b, float f) { return mix3(a, b, f); } };
struct FmtRGBA8 { typedef f3 T; static VKR_DEV T zero() { return mk3(0, 0, 0); }
  static VKR_DEV T decode(uint32_t v) { return mk3(unorm8_to_float(v & 0xFFu), unorm8_to_float((v >> 8) & 0xFFu), unorm8_to_float((v >> 16) & 0xFFu)); }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { return decode(*(const uint32_t*)(t.p + toff(t, lx, ly, 4))); }
  static VKR_DEV T lerp(T a, T b, float f) { return mix3(a, b, f); } };
struct FmtRGBA16U { typedef f4 T; static VKR_DEV T zero() { return mk4(0, 0, 0, 0); }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { uint2 v = *(const uint2*)(t.p + toff(t, lx, ly, 8));
    return mk4(unorm16_to_float(v.x & 0xFFFFu), unorm16_to_float(v.x >> 16), unorm16_to_float(v.y & 0xFFFFu), unorm16_to_float(v.y >> 16)); }
  static VKR_DEV T lerp(T a, T b, float f) { return mix4(a, b, f); } };
struct FmtRGBA16F { typedef f4 T; static VKR_DEV T zero() { return mk4(0, 0, 0, 0); }
  static VKR_DEV T load(const Tex& t, int lx, int ly) { uint2 v = *(const uint2*)(t.p + toff(t, lx, ly, 8));
    return mk4(half_bits_to_float(v.x & 0xFFFFu), half_bits_to_float(v.x >> 16), half_bits_to_float(v.y & 0xFFFFu), half_bits_to_float(v.y >> 16)); }
  static VKR_DEV T lerp(T a, T b, float f) { return mix4(a, b, f); } };

// texelFetch at frame coordinates: outside the frame -> 0; inside the frame but outside
// the window (tiling with too small a halo) -> clamped to the window.
template <class F> VKR_DEV typename F::T fetch(const Tex& t, int gx, int gy) {
  if (gx < 0 || gy < 0 || gx >= t.fw || gy >= t.fh) return F::zero();
  int lx = iclamp(gx - t.ox, 0, t.w - 1), ly = iclamp(gy - t.oy, 0, t.h - 1);
  return F::load(t, lx, ly);
}
// clamp-to-edge fetch of the bilinear filter.  The window lies inside the frame, so clamping to the
// frame and then to the window held in memory is one clamp to the window.
template <class F> VKR_DEV typename F::T fetch_clamped(const Tex& t, int gx, int gy) {
  int lx = iclamp(gx - t.ox, 0, t.w - 1), ly = iclamp(gy - t.oy, 0, t.h - 1);
  return F::load(t, lx, ly);
}
// texture()/textureLod()/textureOffset(): bilinear, clamp-to-edge
template <class F> VKR_DEV typename F::T sample(const Tex& t, f2 uv, int offx = 0, int offy = 0) {
  float x = cfma(uv.x, (float)t.fw, -0.5f), y = cfma(uv.y, (float)t.fh, -0.5f);
  float x0f = floorf(x), y0f = floorf(y);
  float fx = x - x0f, fy = y - y0f;
  int x0 = f2i(x0f) + offx, y0 = f2i(y0f) + offy;
  typename F::T t00 = fetch_clamped<F>(t, x0, y0), t10 = fetch_clamped<F>(t, x0 + 1, y0);
  typename F::T t01 = fetch_clamped<F>(t, x0, y0 + 1), t11 = fetch_clamped<F>(t, x0 + 1, y0 + 1);
  return F::lerp(F::lerp(t00, t10, fx), F::lerp(t01, t11, fx), fy);
}

// ---- sRGB images ------------------------------------------------------------------------------------
// The EOTF table lives in global memory (srgb_tables.inc); kernels that sample _SRGB images copy
// it into LDS once per block (srgb_lut_stage + __syncthreads) so a texel decode is three LDS
// reads instead of three dependent global loads.
#define VKR_SRGB_LUT_SIZE 256
VKR_DEV void srgb_lut_stage(float* lds_lut, int tid, int nthreads) {
  for (int i = tid; i < VKR_SRGB_LUT_SIZE; i += nthreads) lds_lut[i] = __uint_as_float(k_srgb_decode_bits[i]);
}
// linear -> sRGB8 against a threshold table staged in LDS (same binary search as float_to_srgb8)
VKR_DEV void srgb_thresh_stage(float* lds_thresh, int tid, int nthreads) {
  for (int i = tid; i < VKR_SRGB_LUT_SIZE; i += nthreads) lds_thresh[i] = __uint_as_float(k_srgb_thresh_bits[i]);
}
VKR_DEV uint32_t float_to_srgb8_lds(float x, const float* thresh) {
  if (x != x) return 0u;
  int lo = 0, hi = 255;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int mid = (lo + hi + 1) >> 1;
    if (thresh[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return (uint32_t)lo;
}
VKR_DEV uint32_t load_u32_clamped(const Tex& t, int gx, int gy) {
  int lx = iclamp(gx - t.ox, 0, t.w - 1), ly = iclamp(gy - t.oy, 0, t.h - 1);
  return *(const uint32_t*)(t.p + toff(t, lx, ly, 4));
}
// Two / four consecutive dwords from any 4-byte aligned address as ONE global load (dwordx2 / dwordx4 only need dword
// alignment).  The texture-address unit spends the same time on a wave's load whatever its width, so kernels whose
// loads — not their arithmetic — set the pace (TAA) fetch horizontally adjacent texels together.
struct __attribute__((packed, aligned(4))) U32x2 { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) U32x4 { uint32_t x, y, z, w; };
VKR_DEV U32x2 load_u32x2(const uint8_t* p) { return *(const U32x2*)p; }
VKR_DEV U32x4 load_u32x4(const uint8_t* p) { return *(const U32x4*)p; }
struct BilinearTaps { uint32_t t00, t10, t01, t11; float fx, fy; };
// the four raw texels + weights of texture(tex, uv) for any 4-byte format
VKR_DEV BilinearTaps bilinear_taps_u32(const Tex& t, f2 uv) {
  BilinearTaps b;
  float x = cfma(uv.x, (float)t.fw, -0.5f), y = cfma(uv.y, (float)t.fh, -0.5f);
  float x0f = floorf(x), y0f = floorf(y);
  b.fx = x - x0f; b.fy = y - y0f;
  int x0 = f2i(x0f), y0 = f2i(y0f);
  b.t00 = load_u32_clamped(t, x0, y0); b.t10 = load_u32_clamped(t, x0 + 1, y0);
  b.t01 = load_u32_clamped(t, x0, y0 + 1); b.t11 = load_u32_clamped(t, x0 + 1, y0 + 1);
  return b;
}
// The value sample<F>(tex, uv) returns, from taps issued earlier: lets a kernel put the loads of several
// samples in flight together and decode them once they have arrived (one memory latency instead of one each).
template <class F> VKR_DEV typename F::T taps_resolve(const BilinearTaps& b) {
  return F::lerp(F::lerp(F::decode(b.t00), F::decode(b.t10), b.fx), F::lerp(F::decode(b.t01), F::decode(b.t11), b.fx), b.fy);
}
// The bilinear footprint of texture(t, uv [, offset]) on a 4-byte format, each ROW fetched as one 8-byte load of the texel
// pair (xs, xs + 1), xs = clamp(x0, 0, w - 2): the texture-address unit spends the same time on a wave's load whatever
// its width, so two loads instead of four.  left_is_y / right_is_x say which half of the pair the left / right tap is;
// they differ from (x, y) only in the first and last column, where clamp-to-edge makes both taps the same texel.
// Needs w >= 2.  Images with one window geometry and pitch share the record.
struct PairFootprint { uint32_t row0, row1; bool left_is_y, right_is_x; float fx, fy; };
VKR_DEV PairFootprint pair_footprint(const Tex& t, f2 uv, int offx = 0, int offy = 0) {
  PairFootprint f;
  const float x = cfma(uv.x, (float)t.fw, -0.5f), y = cfma(uv.y, (float)t.fh, -0.5f);
  const float x0f = floorf(x), y0f = floorf(y);
  f.fx = x - x0f; f.fy = y - y0f;
  const int x0 = f2i(x0f) + offx - t.ox, y0 = f2i(y0f) + offy - t.oy;
  const int xs = iclamp(x0, 0, t.w - 2);
  f.left_is_y = x0 > xs;    // x0 >= w - 1: both taps are the pair's second texel
  f.right_is_x = x0 < xs;   // x0 <= -1: both taps are the pair's first texel
  const uint32_t xb = (uint32_t)xs * 4u;
  f.row0 = __umul24((uint32_t)iclamp(y0, 0, t.h - 1), (uint32_t)t.pitch) + xb;
  f.row1 = __umul24((uint32_t)iclamp(y0 + 1, 0, t.h - 1), (uint32_t)t.pitch) + xb;
  return f;
}
VKR_DEV BilinearTaps pair_taps(const Tex& t, const PairFootprint& f) {
  BilinearTaps b;
  const U32x2 r0 = load_u32x2(t.p + f.row0), r1 = load_u32x2(t.p + f.row1);
  b.t00 = f.left_is_y ? r0.y : r0.x; b.t10 = f.right_is_x ? r0.x : r0.y;
  b.t01 = f.left_is_y ? r1.y : r1.x; b.t11 = f.right_is_x ? r1.x : r1.y;
  b.fx = f.fx; b.fy = f.fy;
  return b;
}
// one channel (0 = r, 1 = g, 2 = b) of texture() on an RGBA8_SRGB image
VKR_DEV float taps_srgb_channel(const BilinearTaps& b, int channel, const float* lut) {
  const int sh = channel * 8;
  const float a00 = lut[(b.t00 >> sh) & 0xFFu], a10 = lut[(b.t10 >> sh) & 0xFFu];
  const float a01 = lut[(b.t01 >> sh) & 0xFFu], a11 = lut[(b.t11 >> sh) & 0xFFu];
  return mixf(mixf(a00, a10, b.fx), mixf(a01, a11, b.fx), b.fy);
}
VKR_DEV float sample_srgb_channel(const Tex& t, f2 uv, int channel, const float* lut) {
  const BilinearTaps b = bilinear_taps_u32(t, uv);
  const int sh = channel * 8;
  const float a00 = lut[(b.t00 >> sh) & 0xFFu], a10 = lut[(b.t10 >> sh) & 0xFFu];
  const float a01 = lut[(b.t01 >> sh) & 0xFFu], a11 = lut[(b.t11 >> sh) & 0xFFu];
  return mixf(mixf(a00, a10, b.fx), mixf(a01, a11, b.fx), b.fy);
}
VKR_DEV f3 srgb_rgb(uint32_t v, const float* lut) { return mk3(lut[v & 0xFFu], lut[(v >> 8) & 0xFFu], lut[(v >> 16) & 0xFFu]); }
VKR_DEV f3 sample_srgb_rgb(const Tex& t, f2 uv, const float* lut) {
  const BilinearTaps b = bilinear_taps_u32(t, uv);
  return mix3(mix3(srgb_rgb(b.t00, lut), srgb_rgb(b.t10, lut), b.fx), mix3(srgb_rgb(b.t01, lut), srgb_rgb(b.t11, lut), b.fx), b.fy);
}

template <class T> VKR_DEV T* texel_ptr(const Tex& t, int lx, int ly) { return (T*)(const_cast<uint8_t*>(t.p) + toff(t, lx, ly, (int)sizeof(T))); }

// ---- shared shader helpers (gbuffer_encode.glsl / brdf.glsl), literal operation order -------
struct Proj { float tg, aspect, znear, zfar; };  // tg = tanf(fovy/2) evaluated on the host

// gbuffer_encode.glsl:5-7
VKR_DEV float sign_nz(float k) { return (k >= 0.0f) ? 1.0f : -1.0f; }
// gbuffer_encode.glsl:17-27
VKR_DEV f2 encode_normal(f3 v) {
  float l1norm = (fabsf(v.x) + fabsf(v.y)) + fabsf(v.z);
  float inv = 1.0f / l1norm;
  f2 r = mk2(v.x * inv, v.y * inv);
  if (v.z < 0.0f) r = mk2((1.0f - fabsf(r.y)) * sign_nz(r.x), (1.0f - fabsf(r.x)) * sign_nz(r.y));
  return mk2(cfma(0.5f, r.x, 0.5f), cfma(0.5f, r.y, 0.5f));
}
// gbuffer_encode.glsl:29-37
VKR_DEV f3 decode_normal(f2 uv) {
  uv = mk2(cfma(2.0f, uv.x, -1.0f), cfma(2.0f, uv.y, -1.0f));
  f3 v = mk3(uv.x, uv.y, (1.0f - fabsf(uv.x)) - fabsf(uv.y));
  if (v.z < 0.0f) {
    float nx = (1.0f - fabsf(v.y)) * sign_nz(v.x);
    float ny = (1.0f - fabsf(v.x)) * sign_nz(v.y);
    v.x = nx; v.y = ny;
  }
  return normalize(v);
}
// uv of the centre of pixel g of an extent of `size` pixels, (g + 0.5) / size with 0 <= g < size <= 65535: both operands
// and the quotient are normal, so div_normal gives the IEEE quotient (vkr_selftest_pixel_uv checks every pair)
VKR_DEV float pixel_centre_uv(int g, float size) { return div_normal((float)g + 0.5f, size); }
// decode_normal whose result only enters smooth terms (weights, shading angles)
VKR_DEV f3 decode_normal_fast(f2 uv) {
  uv = mk2(cfma(2.0f, uv.x, -1.0f), cfma(2.0f, uv.y, -1.0f));
  f3 v = mk3(uv.x, uv.y, (1.0f - fabsf(uv.x)) - fabsf(uv.y));
  if (v.z < 0.0f) {
    float nx = (1.0f - fabsf(v.y)) * sign_nz(v.x);
    float ny = (1.0f - fabsf(v.x)) * sign_nz(v.y);
    v.x = nx; v.y = ny;
  }
  return normalize_fast(v);
}
// gbuffer_encode.glsl:53-56
VKR_DEV float linearize_depth2(float d, float n, float f) { return (n * f) / cfma(d, f - n, -f); }
// the same for a stored depth d in [0,1]: the denominator lies in [-f, -n], far inside the normal range
VKR_DEV float linearize_depth2_unorm(float d, float n, float f) { return div_normal(n * f, cfma(d, f - n, -f)); }
// gbuffer_encode.glsl:58-69.  d is always a depth-buffer value (or a lerp of them) in [0,1] here.
VKR_DEV f3 reconstruct_view_vec(f2 uv, float d, const Proj& pr) {
  float z = linearize_depth2_unorm(d, pr.znear, pr.zfar);
  float xd = cfma(2.0f, uv.x, -1.0f), yd = cfma(2.0f, uv.y, -1.0f);
  float x = -(xd) * ((z * pr.aspect) * pr.tg);
  float y = -(yd) * (z * pr.tg);
  return mk3(x, y, z);
}
// gbuffer_encode.glsl:71-73
VKR_DEV float encode_depth(float z, float n, float f) { return f / (f - n) + (f * n) / (z * (f - n)); }
// gbuffer_encode.glsl:75-84
// f_over_fn: zfar / (zfar - znear), the one quotient of the expression that is the same for every pixel (the trace gets
// it from the host, where the same IEEE division produces the same float)
VKR_DEV f3 project_view_vec(f3 v, const Proj& pr, float f_over_fn) {
  float n = pr.znear, f = pr.zfar, z = v.z;
  float depth = f_over_fn + (f * n) / (z * (f - n));
  float pu = v.x / ((-v.z * pr.tg) * pr.aspect);
  float pv = v.y / (-z * pr.tg);
  return mk3(cfma(0.5f, pu, 0.5f), cfma(0.5f, pv, 0.5f), depth);
}
VKR_DEV f3 project_view_vec(f3 v, const Proj& pr) {
  float n = pr.znear, f = pr.zfar, z = v.z;
  float depth = f / (f - n) + (f * n) / (z * (f - n));
  float pu = v.x / ((-v.z * pr.tg) * pr.aspect);
  float pv = v.y / (-z * pr.tg);
  return mk3(cfma(0.5f, pu, 0.5f), cfma(0.5f, pv, 0.5f), depth);
}

// brdf.glsl:6-13
VKR_DEV f3 fresnelSchlick(float cos_theta, f3 F0) {
  float p = powf(vclamp(1.0f - cos_theta, 0.0f, 1.0f), 5.0f);
  return F0 + (mk3(1.0f, 1.0f, 1.0f) - F0) * p;
}
VKR_DEV f3 F0_approximation(f3 albedo, float metallic) { return mix3(mk3(0.04f, 0.04f, 0.04f), albedo, metallic); }
// brdf.glsl:43-56
VKR_DEV float brdfG1(float alpha2, float NdotV) {
  float NdotV2 = NdotV * NdotV;
  float tgv2 = (1.0f - NdotV2) / NdotV2;
  return 2.0f / (1.0f + sqrtf(1.0f + alpha2 * tgv2));
}
VKR_DEV float brdfG2(float NdotV, float NdotL, float alpha2) {
  float NdotV2 = NdotV * NdotV, NdotL2 = NdotL * NdotL;
  float L1 = sqrtf(1.0f + (alpha2 * (1.0f - NdotV2)) / NdotV2);
  float L2 = sqrtf(1.0f + (alpha2 * (1.0f - NdotL2)) / NdotL2);
  return 2.0f / (L1 + L2);
}
// brdf.glsl:107-128 (live #else branch)
VKR_DEV float sampleGGXdirPDF(const Tex& pdf_tex, f3 V, f3 N, f3 L, float alpha) {
  f3 Y = normalize(cross(V, N));
  f3 X = normalize(cross(Y, V));
  alpha = vclamp(alpha, 0.0f, 0.9f);
  f3 Lproj = normalize(L - V * dot(V, L));
  float cos_theta = dot(X, Lproj);
  const float cos_phin = dot(N, X);
  const float sin_phin = sqrt_ieee(1.0f - cos_phin * cos_phin);
  const float alpha2 = alpha * alpha;
  const float coef = sqrt_ieee(1.0f - alpha2);
  const float a = ((0.5f * coef) * cos_phin) * cos_theta + 0.5f;
  const float b = coef * sin_phin;
  return alpha2 / ((2.0f * VKR_PI) * coef) * sample<FmtR32F>(pdf_tex, mk2(a, b));
}

// (float)sin((double)x) for the argument of the shaders' rand() hash (trace.comp:156-158: x = dot(uv, (12.9898, 78.233)),
// 0 <= x < 92; valid for |x| < 1e5).  The contract evaluates that sine in double and rounds once, because its low bits pick the
// Halton entry.  The generic double sin of the device library spends several hundred instructions (it also handles huge
// arguments); here: Cody-Waite reduction by pi/2 in two exact steps (k < 2^17 keeps k * PIO2_HI exact: 33 significant
// bits) and the fdlibm kernel polynomials on |r| <= pi/4, ~25 double operations, < 1 ulp in double — rounding that to
// float gives the same float as any other < 1-ulp double sine except on a ~1e-9 sliver around float rounding ties.
VKR_DEV float sin_hash_arg(float x) {
  const double xd = (double)x;
  const double kd = __builtin_rint(xd * 0.63661977236758138243);  // 2 / pi
  const int k = (int)kd;
  double r = __builtin_fma(-kd, 1.57079632673412561417e+00, xd);  // pi/2, first 33 bits
  r = __builtin_fma(-kd, 6.07710050650619224932e-11, r);          // pi/2 - the above
  const double z = r * r;
  // __kernel_sin: r + r^3 (S1 + z (S2 + z (S3 + z (S4 + z (S5 + z S6)))))
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double sn = __builtin_fma(z * r, ps, r);
  // __kernel_cos: 1 - z/2 + z^2 (C1 + z (C2 + z (C3 + z (C4 + z (C5 + z C6)))))
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const double v = (k & 1) ? cs : sn;
  return (float)((k & 2) ? -v : v);
}

// the cosine-weighted horizon arc shared by main.comp:246-248 and trace.comp:127-134
VKR_DEV float arc_occlusion(float h, float n, float len_np) {
  return (((1.0f / VKR_PI) * len_np) * 0.25f) * vmax((-cosf(2.0f * h - n) + cosf(n)) + (2.0f * h) * sinf(n), 0.0f);
}

}  // namespace vkr
