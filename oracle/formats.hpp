// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// formats.hpp — storage-format codecs and the sampling semantics of
// gpu::DEFAULT_SAMPLER (reference src/gpu/samplers.hpp:36-55): bilinear min/mag,
// clamp-to-edge, also on depth; texelFetch out of bounds (or beyond the view's last
// mip) returns 0; storage-image stores use the host-created
// format (Appendix A.5) with round-to-nearest-even.
#pragma once
#include "glsl.hpp"
#include "../include/vkr_postfx.h"

namespace oracle {
using namespace glsl;

// ---- scalar codecs ---------------------------------------------------------------
inline float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: normalise
      int e = -1;
      do { e++; man <<= 1; } while ((man & 0x400u) == 0);
      bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3FFu) << 13;
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | man << 13;
  } else {
    bits = sign | (exp + 112u) << 23 | man << 13;
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// fp32 -> fp16, round to nearest even, overflow -> inf, NaN -> quiet NaN (v_cvt_f16_f32)
inline uint16_t float_to_half(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7FFFFFFFu;
  if (ax > 0x7F800000u) return (uint16_t)(sign | 0x7E00u | ((ax >> 13) & 0x1FFu));  // NaN (keep payload top bits, quiet)
  if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // >= 65520 -> inf
  if (ax < 0x38800000u) {                                    // < 2^-14: subnormal half
    if (ax < 0x33000000u) return (uint16_t)sign;             // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
    uint32_t e = ax >> 23;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
    uint32_t shift = 126u - e;  // 14..24
    uint32_t r = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1u);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ax - 0x38000000u;  // rebias
  uint32_t rem = r & 0x1FFFu;
  r >>= 13;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}

inline float d24_to_float(uint32_t texel) { return (float)(texel & 0xFFFFFFu) / 16777215.0f; }
inline uint32_t float_to_d24(float d) { return (uint32_t)rintf(clamp(d, 0.0f, 1.0f) * 16777215.0f) & 0xFFFFFFu; }
inline float unorm16_to_float(uint16_t v) { return (float)v / 65535.0f; }
inline uint16_t float_to_unorm16(float f) { return (uint16_t)rintf(clamp(f, 0.0f, 1.0f) * 65535.0f); }
inline float unorm8_to_float(uint8_t v) { return (float)v / 255.0f; }
inline uint8_t float_to_unorm8(float f) { return (uint8_t)rintf(clamp(f, 0.0f, 1.0f) * 255.0f); }

// sRGB EOTF as a 256-entry table (IEC 61966-2-1), values rounded once from double.
struct SrgbTables {
  float decode[256];
  float thresh[256];  // thresh[i] = linear value above which code >= i  (midpoints in encoded space)
  SrgbTables() {
    for (int i = 0; i < 256; i++) {
      double c = i / 255.0;
      decode[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    thresh[0] = -1.0f;
    for (int i = 1; i < 256; i++) {
      double c = (i - 0.5) / 255.0;
      thresh[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
  }
};
inline const SrgbTables& srgb() { static SrgbTables t; return t; }
inline float srgb8_to_float(uint8_t v) { return srgb().decode[v]; }
// encode = number of thresholds <= x, exact and monotone (no pow at run time)
inline uint8_t float_to_srgb8(float x) {
  const float* t = srgb().thresh;
  int lo = 0, hi = 255;  // largest i with t[i] <= x
  if (!(x == x)) return 0;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (t[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return (uint8_t)lo;
}

// ---- accounting of the frozen undefined behaviour ------------------------------
// A texelFetch outside the frame or beyond the view's last mip is undefined in the reference (no robust image access,
// gpu/driver.cpp:185-189); the restatement freezes "returns 0".  To bound what that choice is worth, Image::fetch marks
// the calling thread when the rule produced the value, a UbPixel guard around one output pixel turns the mark into a
// per-pass count, and ub_oob_mode switches the rule to the other plausible hardware behaviour (clamp to the edge texel /
// the last mip) so that a second run shows which outputs actually change.  tools/ub_masks.py drives this.
enum UbPass { UB_SSSR_TRACE = 0, UB_SSSR_FILTER, UB_SSSR_BLUR, UB_GTAO_MAIN, UB_GTAO_FILTER, UB_GTAO_ACCUMULATE, UB_TAA, UB_PASS_COUNT };
enum { UB_OUT_OF_FRAME = 1u, UB_BEYOND_LAST_MIP = 2u };
extern thread_local uint32_t ub_flags;
extern int ub_oob_mode;  // 0: frozen rule (fetch returns 0); 1: clamp to the frame edge / last mip (sensitivity runs only)
void ub_count(int pass, uint32_t flags);
struct UbPixel {
  int pass;
  explicit UbPixel(int p) : pass(p) { ub_flags = 0; }
  ~UbPixel() { if (ub_flags) ub_count(pass, ub_flags); }
};

// ---- image view --------------------------------------------------------------------
struct Image {
  vkr_img d;
  explicit Image(const vkr_img& v) : d(v) {}

  int mips() const { return (int)d.mip_count; }
  int w(int mip = 0) const { int v = (int)(d.width >> mip); return v > 0 ? v : 1; }
  int h(int mip = 0) const { int v = (int)(d.height >> mip); return v > 0 ? v : 1; }
  int fw(int mip = 0) const { int v = (int)(d.full_width >> mip); return v > 0 ? v : 1; }
  int fh(int mip = 0) const { int v = (int)(d.full_height >> mip); return v > 0 ? v : 1; }
  int ox(int mip = 0) const { return d.origin_x >> mip; }
  int oy(int mip = 0) const { return d.origin_y >> mip; }
  // textureSize()/imageSize(): the whole frame's extent (what the shader sees)
  ivec2 size(int mip = 0) const { return ivec2(fw(mip), fh(mip)); }

  uint8_t* texel_ptr(int lx, int ly, int mip) const {
    return (uint8_t*)d.base + d.mip_offset[mip] + (size_t)ly * d.pitch_bytes[mip] +
           (size_t)lx * vkr_format_bytes(d.format);
  }

  // decode the texel at *local* coordinates (must be in range)
  vec4 load_local(int lx, int ly, int mip) const {
    const uint8_t* p = texel_ptr(lx, ly, mip);
    switch (d.format) {
      case VKR_FMT_D24_UNORM_S8: { uint32_t t; std::memcpy(&t, p, 4); return vec4(d24_to_float(t), 0, 0, 1); }
      case VKR_FMT_RG16_UNORM: { uint16_t t[2]; std::memcpy(t, p, 4); return vec4(unorm16_to_float(t[0]), unorm16_to_float(t[1]), 0, 1); }
      case VKR_FMT_RG16_SFLOAT: { uint16_t t[2]; std::memcpy(t, p, 4); return vec4(half_to_float(t[0]), half_to_float(t[1]), 0, 1); }
      case VKR_FMT_RGBA8_SRGB: return vec4(srgb8_to_float(p[0]), srgb8_to_float(p[1]), srgb8_to_float(p[2]), unorm8_to_float(p[3]));
      case VKR_FMT_RGBA8_UNORM: return vec4(unorm8_to_float(p[0]), unorm8_to_float(p[1]), unorm8_to_float(p[2]), unorm8_to_float(p[3]));
      case VKR_FMT_RGBA16_UNORM: { uint16_t t[4]; std::memcpy(t, p, 8); return vec4(unorm16_to_float(t[0]), unorm16_to_float(t[1]), unorm16_to_float(t[2]), unorm16_to_float(t[3])); }
      case VKR_FMT_RGBA16_SFLOAT: { uint16_t t[4]; std::memcpy(t, p, 8); return vec4(half_to_float(t[0]), half_to_float(t[1]), half_to_float(t[2]), half_to_float(t[3])); }
      case VKR_FMT_R16_SFLOAT: { uint16_t t; std::memcpy(&t, p, 2); return vec4(half_to_float(t), 0, 0, 1); }
      case VKR_FMT_R32_SFLOAT: { float t; std::memcpy(&t, p, 4); return vec4(t, 0, 0, 1); }
      case VKR_FMT_R8_UNORM: return vec4(unorm8_to_float(p[0]), 0, 0, 1);
      default: return vec4();
    }
  }

  // texelFetch(tex, ivec2(gx,gy), mip): frame coordinates; out of frame, out of the
  // mip range -> 0.  Inside the frame but outside the window held in memory (only
  // possible when tiled with too small a halo): clamped to the window.
  vec4 fetch(int gx, int gy, int mip) const {
    if (mip < 0 || mip >= mips()) {
      ub_flags |= UB_BEYOND_LAST_MIP;
      if (ub_oob_mode == 0 || mip < 0) return vec4();
      // sensitivity run: the coordinate was computed for `mip`; rescale it to the last mip that exists
      const int last = mips() - 1;
      return fetch_clamped(gx << (mip - last), gy << (mip - last), last);
    }
    if (gx < 0 || gy < 0 || gx >= fw(mip) || gy >= fh(mip)) {
      ub_flags |= UB_OUT_OF_FRAME;
      if (ub_oob_mode == 0) return vec4();
      return fetch_clamped(gx, gy, mip);
    }
    int lx = clamp(gx - ox(mip), 0, w(mip) - 1);
    int ly = clamp(gy - oy(mip), 0, h(mip) - 1);
    return load_local(lx, ly, mip);
  }
  vec4 fetch(ivec2 p, int mip) const { return fetch(p.x, p.y, mip); }

  // clamp-to-edge fetch used by the bilinear filter
  vec4 fetch_clamped(int gx, int gy, int mip) const {
    gx = clamp(gx, 0, fw(mip) - 1);
    gy = clamp(gy, 0, fh(mip) - 1);
    int lx = clamp(gx - ox(mip), 0, w(mip) - 1);
    int ly = clamp(gy - oy(mip), 0, h(mip) - 1);
    return load_local(lx, ly, mip);
  }

  // texture()/textureLod() with an integral lod, optional textureOffset texel offset
  vec4 sample(vec2 uv, int mip = 0, ivec2 offset = ivec2(0, 0)) const {
    if (mip >= mips()) mip = mips() - 1;
    float x = cfma(uv.x, (float)fw(mip), -0.5f);
    float y = cfma(uv.y, (float)fh(mip), -0.5f);
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = f2i(x0f) + offset.x, y0 = f2i(y0f) + offset.y;
    vec4 t00 = fetch_clamped(x0, y0, mip), t10 = fetch_clamped(x0 + 1, y0, mip);
    vec4 t01 = fetch_clamped(x0, y0 + 1, mip), t11 = fetch_clamped(x0 + 1, y0 + 1, mip);
    return mix(mix(t00, t10, fx), mix(t01, t11, fx), fy);
  }

  // imageStore with the image's real format; (gx,gy) in frame coordinates, must lie in the window
  void store(int gx, int gy, vec4 v, int mip = 0) const {
    int lx = gx - ox(mip), ly = gy - oy(mip);
    if (lx < 0 || ly < 0 || lx >= w(mip) || ly >= h(mip)) return;
    uint8_t* p = texel_ptr(lx, ly, mip);
    switch (d.format) {
      case VKR_FMT_D24_UNORM_S8: { uint32_t t = float_to_d24(v.x); std::memcpy(p, &t, 4); break; }
      case VKR_FMT_RG16_UNORM: { uint16_t t[2] = {float_to_unorm16(v.x), float_to_unorm16(v.y)}; std::memcpy(p, t, 4); break; }
      case VKR_FMT_RG16_SFLOAT: { uint16_t t[2] = {float_to_half(v.x), float_to_half(v.y)}; std::memcpy(p, t, 4); break; }
      case VKR_FMT_RGBA8_SRGB: { p[0] = float_to_srgb8(v.x); p[1] = float_to_srgb8(v.y); p[2] = float_to_srgb8(v.z); p[3] = float_to_unorm8(v.w); break; }
      case VKR_FMT_RGBA8_UNORM: { p[0] = float_to_unorm8(v.x); p[1] = float_to_unorm8(v.y); p[2] = float_to_unorm8(v.z); p[3] = float_to_unorm8(v.w); break; }
      case VKR_FMT_RGBA16_UNORM: { uint16_t t[4] = {float_to_unorm16(v.x), float_to_unorm16(v.y), float_to_unorm16(v.z), float_to_unorm16(v.w)}; std::memcpy(p, t, 8); break; }
      case VKR_FMT_RGBA16_SFLOAT: { uint16_t t[4] = {float_to_half(v.x), float_to_half(v.y), float_to_half(v.z), float_to_half(v.w)}; std::memcpy(p, t, 8); break; }
      case VKR_FMT_R16_SFLOAT: { uint16_t t = float_to_half(v.x); std::memcpy(p, &t, 2); break; }
      case VKR_FMT_R32_SFLOAT: { std::memcpy(p, &v.x, 4); break; }
      case VKR_FMT_R8_UNORM: { p[0] = float_to_unorm8(v.x); break; }
      default: break;
    }
  }
  // raw 32-bit texel access (depth integer path of the Hi-Z build)
  uint32_t load_u32(int lx, int ly, int mip) const { uint32_t t; std::memcpy(&t, texel_ptr(lx, ly, mip), 4); return t; }
  void store_u32(int lx, int ly, int mip, uint32_t t) const { std::memcpy(texel_ptr(lx, ly, mip), &t, 4); }
};

}  // namespace oracle
