// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// api.cpp — helpers exported next to the vkr_ref_* pass entries so Python tests can
// probe the restated helper functions directly (known-answer tests of SURVEY.md 8(c)).
#include "shader_common.hpp"
#include <omp.h>

using namespace oracle;

namespace oracle {
thread_local uint32_t ub_flags = 0;
int ub_oob_mode = 0;
static uint64_t g_ub_counts[UB_PASS_COUNT][2];  // [pass][0: out of frame, 1: beyond the last mip] output pixels
void ub_count(int pass, uint32_t flags) {
  if (flags & UB_OUT_OF_FRAME) __atomic_fetch_add(&g_ub_counts[pass][0], 1, __ATOMIC_RELAXED);
  if (flags & UB_BEYOND_LAST_MIP) __atomic_fetch_add(&g_ub_counts[pass][1], 1, __ATOMIC_RELAXED);
}
}  // namespace oracle

extern "C" {

// undefined-behaviour accounting (formats.hpp): counts of output pixels whose value went through the frozen OOB rule
void vkr_ref_ub_reset(void) { std::memset(g_ub_counts, 0, sizeof(g_ub_counts)); }
void vkr_ref_ub_counts(uint64_t* out /* [UB_PASS_COUNT][2] */) { std::memcpy(out, g_ub_counts, sizeof(g_ub_counts)); }
void vkr_ref_ub_set_oob_mode(int mode) { ub_oob_mode = mode; }

uint32_t vkr_format_bytes(uint32_t format) {
  switch (format) {
    case VKR_FMT_D24_UNORM_S8: case VKR_FMT_RG16_UNORM: case VKR_FMT_RG16_SFLOAT:
    case VKR_FMT_RGBA8_SRGB: case VKR_FMT_RGBA8_UNORM: case VKR_FMT_R32_SFLOAT: return 4;
    case VKR_FMT_RGBA16_UNORM: case VKR_FMT_RGBA16_SFLOAT: return 8;
    case VKR_FMT_RGBA32_SFLOAT: return 16;
    case VKR_FMT_R16_SFLOAT: return 2;
    case VKR_FMT_R8_UNORM: return 1;
    default: return 0;
  }
}

uint32_t vkr_ref_numeric_contract(void) { return VKR_CONTRACT; }
int vkr_ref_threads(void) { return omp_get_max_threads(); }
void vkr_ref_set_threads(int n) { omp_set_num_threads(n); }

void vkr_ref_encode_normal(const float* n3, float* out2) { vec2 e = encode_normal(vec3(n3[0], n3[1], n3[2])); out2[0] = e.x; out2[1] = e.y; }
void vkr_ref_decode_normal(const float* uv2, float* out3) { vec3 n = decode_normal(vec2(uv2[0], uv2[1])); out3[0] = n.x; out3[1] = n.y; out3[2] = n.z; }
float vkr_ref_linearize_depth2(float d, float n, float f) { return linearize_depth2(d, n, f); }
float vkr_ref_encode_depth(float z, float n, float f) { return encode_depth(z, n, f); }
void vkr_ref_reconstruct_view_vec(const float* uv2, float d, float fovy, float aspect, float n, float f, float* out3) {
  vec3 v = reconstruct_view_vec(vec2(uv2[0], uv2[1]), d, fovy, aspect, n, f); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
void vkr_ref_project_view_vec(const float* v3, float fovy, float aspect, float n, float f, float* out3) {
  vec3 v = project_view_vec(vec3(v3[0], v3[1], v3[2]), fovy, aspect, n, f); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
uint16_t vkr_ref_float_to_half(float f) { return float_to_half(f); }
float vkr_ref_half_to_float(uint16_t h) { return half_to_float(h); }
uint8_t vkr_ref_float_to_srgb8(float f) { return float_to_srgb8(f); }
float vkr_ref_srgb8_to_float(uint8_t v) { return srgb8_to_float(v); }
float vkr_ref_brdfG1(float alpha2, float ndv) { return brdfG1(alpha2, ndv); }
float vkr_ref_brdfG2(float ndv, float ndl, float alpha2) { return brdfG2(ndv, ndl, alpha2); }

// bilinear sample / texelFetch of an arbitrary image (sampler known-answer tests)
void vkr_ref_sample(const vkr_img* img, float u, float v, int mip, int offx, int offy, float* out4) {
  vec4 t = Image(*img).sample(vec2(u, v), mip, ivec2(offx, offy)); out4[0] = t.x; out4[1] = t.y; out4[2] = t.z; out4[3] = t.w;
}
void vkr_ref_fetch(const vkr_img* img, int x, int y, int mip, float* out4) {
  vec4 t = Image(*img).fetch(x, y, mip); out4[0] = t.x; out4[1] = t.y; out4[2] = t.z; out4[3] = t.w;
}
// generic hierarchical march (screen_trace.glsl:51-100) on one ray
int vkr_ref_hierarchical_raymarch(const vkr_img* depth, const float* origin3, const float* dir3, int most_detailed_mip,
                                  uint32_t max_steps, float* out_pos3) {
  bool valid = false;
  vec3 p = hierarchical_raymarch(Image(*depth), vec3(origin3[0], origin3[1], origin3[2]), vec3(dir3[0], dir3[1], dir3[2]),
                                 most_detailed_mip, max_steps, valid);
  out_pos3[0] = p.x; out_pos3[1] = p.y; out_pos3[2] = p.z;
  return valid ? 1 : 0;
}

}  // extern "C"

// gtao_direction (main.comp:276-278) for the known-answer table of SURVEY.md 8(c)(4)
extern "C" float vkr_ref_gtao_direction(int x, int y) { return (1.0f / 16.0f) * (float)((((x + y) & 3) << 2) + (x & 3)); }
