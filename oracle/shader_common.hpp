// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// shader_common.hpp — CPU restatement of the reference's shared GLSL helpers:
//   src/shaders/include/gbuffer_encode.glsl, brdf.glsl, screen_trace.glsl.
// Every function cites the lines it follows.  Operation order is literal.
#pragma once
#include "formats.hpp"

namespace oracle {

// gbuffer_encode.glsl:5-15
inline float sign_nz(float k) { return (k >= 0.0f) ? 1.0f : -1.0f; }

// gbuffer_encode.glsl:17-27
inline vec2 encode_normal(vec3 v) {
  float l1norm = (abs(v.x) + abs(v.y)) + abs(v.z);
  float inv = 1.0f / l1norm;
  vec2 result(v.x * inv, v.y * inv);
  if (v.z < 0.0f) {
    vec2 r2((1.0f - abs(result.y)) * sign_nz(result.x), (1.0f - abs(result.x)) * sign_nz(result.y));
    result = r2;
  }
  return vec2(cfma(0.5f, result.x, 0.5f), cfma(0.5f, result.y, 0.5f));
}

// gbuffer_encode.glsl:29-37
inline vec3 decode_normal(vec2 uv) {
  uv = vec2(cfma(2.0f, uv.x, -1.0f), cfma(2.0f, uv.y, -1.0f));
  vec3 v(uv.x, uv.y, (1.0f - abs(uv.x)) - abs(uv.y));
  if (v.z < 0.0f) {
    float nx = (1.0f - abs(v.y)) * sign_nz(v.x);
    float ny = (1.0f - abs(v.x)) * sign_nz(v.y);
    v.x = nx;
    v.y = ny;
  }
  return normalize(v);
}

// gbuffer_encode.glsl:39-43
inline vec3 sample_gbuffer_normal(const Image& normal_tex, vec2 uv) {
  vec4 t = normal_tex.sample(uv);
  return decode_normal(t.xy());
}

// gbuffer_encode.glsl:53-56
inline float linearize_depth2(float d, float n, float f) { return (n * f) / cfma(d, f - n, -f); }

// gbuffer_encode.glsl:58-69.  tan(fovy/2) is evaluated with the host libm (tanf).
inline vec3 reconstruct_view_vec(vec2 uv, float d, float fovy, float aspect, float z_near, float z_far) {
  float tg_alpha = tanf(fovy / 2.0f);
  float z = linearize_depth2(d, z_near, z_far);
  float xd = cfma(2.0f, uv.x, -1.0f);
  float yd = cfma(2.0f, uv.y, -1.0f);
  float x = -(xd) * ((z * aspect) * tg_alpha);
  float y = -(yd) * (z * tg_alpha);
  return vec3(x, y, z);
}

// gbuffer_encode.glsl:71-73
inline float encode_depth(float z, float n, float f) { return f / (f - n) + (f * n) / (z * (f - n)); }

// gbuffer_encode.glsl:75-84
inline vec3 project_view_vec(vec3 v, float fovy, float aspect, float n, float f) {
  float tg_alpha = tanf(fovy / 2.0f);
  float z = v.z;
  float depth = f / (f - n) + (f * n) / (z * (f - n));
  float pu = v.x / ((-v.z * tg_alpha) * aspect);
  float pv = v.y / (-z * tg_alpha);
  return vec3(cfma(0.5f, pu, 0.5f), cfma(0.5f, pv, 0.5f), depth);
}

// ---- brdf.glsl ---------------------------------------------------------------------
// brdf.glsl:6-8
inline vec3 fresnelSchlick(float cos_theta, vec3 F0) {
  float p = powf(clamp(1.0f - cos_theta, 0.0f, 1.0f), 5.0f);
  return F0 + (vec3(1.0f) - F0) * p;
}
// brdf.glsl:10-13
inline vec3 F0_approximation(vec3 albedo, float metallic) { return mix(vec3(0.04f), albedo, metallic); }

// brdf.glsl:31-38 (the #else branch is the live one)
inline float DistributionGGX(vec3 N, vec3 H, float alpha) {
  float NoH = dot(N, H);
  float alpha2 = alpha * alpha;
  float NoH2 = NoH * NoH;
  float den = NoH2 * alpha2 + (1.0f - NoH2);
  return (((NoH2 > 0.0f) ? 1.0f : 0.0f) * alpha2) / ((PI * den) * den);
}

// brdf.glsl:43-47
inline float brdfG1(float alpha2, float NdotV) {
  float NdotV2 = NdotV * NdotV;
  float tgv2 = (1.0f - NdotV2) / NdotV2;
  return 2.0f / (1.0f + sqrtf(1.0f + alpha2 * tgv2));
}
// brdf.glsl:49-56
inline float brdfG2(float NdotV, float NdotL, float alpha2) {
  float NdotV2 = NdotV * NdotV;
  float NdotL2 = NdotL * NdotL;
  float L1 = sqrtf(1.0f + (alpha2 * (1.0f - NdotV2)) / NdotV2);
  float L2 = sqrtf(1.0f + (alpha2 * (1.0f - NdotL2)) / NdotL2);
  return 2.0f / (L1 + L2);
}

// brdf.glsl:107-128 (the #else branch)
inline float sampleGGXdirPDF(const Image& PDF_TEX, vec3 V, vec3 N, vec3 L, float alpha) {
  vec3 Y = normalize(cross(V, N));
  vec3 X = normalize(cross(Y, V));
  alpha = clamp(alpha, 0.0f, 0.9f);
  vec3 Lproj = normalize(L - V * dot(V, L));
  float cos_theta = dot(X, Lproj);
  const float cos_phin = dot(N, X);
  const float sin_phin = sqrtf(1.0f - cos_phin * cos_phin);
  const float alpha2 = alpha * alpha;
  const float coef = sqrtf(1.0f - alpha2);
  const float a = ((0.5f * coef) * cos_phin) * cos_theta + 0.5f;
  const float b = coef * sin_phin;
  float pdf = alpha2 / ((2.0f * PI) * coef) * PDF_TEX.sample(vec2(a, b)).x;
  return pdf;
}

// brdf.glsl:135-155.  cos/sin of phi are evaluated in double and rounded once, so a
// GPU implementation doing the same agrees bit-for-bit (they steer the ray march).
inline vec3 sampleGGXVNDF(vec3 Ve, float alpha_x, float alpha_y, float U1, float U2) {
  vec3 Vh = normalize(vec3(alpha_x * Ve.x, alpha_y * Ve.y, Ve.z));
  float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
  vec3 T1 = lensq > 0.0f ? vec3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : vec3(1, 0, 0);
  vec3 T2 = cross(Vh, T1);
  float r = sqrtf(U1);
  float phi = (2.0f * PI) * U2;
  float t1 = r * (float)std::cos((double)phi);
  float t2 = r * (float)std::sin((double)phi);
  float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
  vec3 Nh = (t1 * T1 + t2 * T2) + sqrtf(max(0.0f, (1.0f - t1 * t1) - t2 * t2)) * Vh;
  vec3 Ne = normalize(vec3(alpha_x * Nh.x, alpha_y * Nh.y, max(0.0f, Nh.z)));
  return Ne;
}

// ---- screen_trace.glsl -------------------------------------------------------------
static const float MAX_T_FLOAT = 3.402823466e+38f;  // screen_trace.glsl:4

// screen_trace.glsl:8-15
inline void initial_advance_ray(vec3 origin, vec3 dir, vec3 inv_dir, vec2 mip_res, vec2 inv_mip_res,
                                vec2 floor_offset, vec2 uv_offset, vec3& pos, float& current_t) {
  vec2 cur_pos = mip_res * origin.xy();
  vec2 xy_plane = floor(cur_pos) + floor_offset;
  xy_plane = vec2(cfma(xy_plane.x, inv_mip_res.x, uv_offset.x), cfma(xy_plane.y, inv_mip_res.y, uv_offset.y));
  vec2 t = (xy_plane - origin.xy()) * inv_dir.xy();
  current_t = min(t.x, t.y);
  pos = madd(origin, current_t, dir);
}

// screen_trace.glsl:17-45
inline bool advance_ray(vec3 origin, vec3 direction, vec3 inv_direction, vec2 current_mip_position,
                        vec2 current_mip_resolution_inv, vec2 floor_offset, vec2 uv_offset, float surface_z,
                        vec3& position, float& current_t) {
  vec2 xy_plane = floor(current_mip_position) + floor_offset;
  xy_plane = vec2(cfma(xy_plane.x, current_mip_resolution_inv.x, uv_offset.x), cfma(xy_plane.y, current_mip_resolution_inv.y, uv_offset.y));
  vec3 boundary_planes(xy_plane, surface_z);
  vec3 t = (boundary_planes - origin) * inv_direction;
  t.z = direction.z > 0.0f ? t.z : MAX_T_FLOAT;
  float t_min = min(min(t.x, t.y), t.z);
  bool above_surface = surface_z > position.z;
  bool skipped_tile = t_min != t.z && above_surface;
  current_t = above_surface ? t_min : current_t;
  position = madd(origin, current_t, direction);
  return skipped_tile;
}

// screen_trace.glsl:47-49; pow(0.5, mip) is an exact power of two
inline vec2 get_mip_resolution(vec2 screen_dimensions, int mip_level) {
  return screen_dimensions * ldexpf(1.0f, -mip_level);
}

struct MarchSetup {
  vec3 inv_direction;
  vec2 uv_offset, floor_offset, res, res_inv;
};
// screen_trace.glsl:54-76 == trace.comp:209-231 (shared prologue of both marches)
inline MarchSetup march_setup(const Image& depth_tex, vec3 direction, int most_detailed_mip) {
  MarchSetup s;
  s.inv_direction = vec3(direction.x != 0.0f ? 1.0f / direction.x : MAX_T_FLOAT,
                         direction.y != 0.0f ? 1.0f / direction.y : MAX_T_FLOAT,
                         direction.z != 0.0f ? 1.0f / direction.z : MAX_T_FLOAT);
  ivec2 ts = depth_tex.size(0);
  vec2 screen_size((float)ts.x, (float)ts.y);
  s.res = get_mip_resolution(screen_size, most_detailed_mip);
  s.res_inv = vec2(1.0f / s.res.x, 1.0f / s.res.y);
  vec2 uvo = (0.005f * ldexpf(1.0f, most_detailed_mip)) / screen_size;
  s.uv_offset = vec2(direction.x < 0.0f ? -uvo.x : uvo.x, direction.y < 0.0f ? -uvo.y : uvo.y);
  s.floor_offset = vec2(direction.x < 0.0f ? 0.0f : 1.0f, direction.y < 0.0f ? 0.0f : 1.0f);
  return s;
}

// screen_trace.glsl:51-100
inline vec3 hierarchical_raymarch(const Image& depth_tex, vec3 origin, vec3 direction, int most_detailed_mip,
                                  uint32_t max_traversal_intersections, bool& valid_hit) {
  MarchSetup s = march_setup(depth_tex, direction, most_detailed_mip);
  int current_mip = most_detailed_mip;
  vec2 current_mip_resolution = s.res;
  vec2 current_mip_resolution_inv = s.res_inv;
  float current_t;
  vec3 position;
  initial_advance_ray(origin, direction, s.inv_direction, current_mip_resolution, current_mip_resolution_inv,
                      s.floor_offset, s.uv_offset, position, current_t);
  uint32_t i = 0;
  while (i < max_traversal_intersections && current_mip >= most_detailed_mip) {
    vec2 current_mip_position = current_mip_resolution * position.xy();
    float surface_z = depth_tex.fetch(to_ivec2(current_mip_position), current_mip).x;
    bool skipped_tile = advance_ray(origin, direction, s.inv_direction, current_mip_position,
                                    current_mip_resolution_inv, s.floor_offset, s.uv_offset, surface_z,
                                    position, current_t);
    current_mip += skipped_tile ? 1 : -1;
    current_mip_resolution *= skipped_tile ? 0.5f : 2.0f;
    current_mip_resolution_inv *= skipped_tile ? 2.0f : 0.5f;
    ++i;
  }
  valid_hit = (i <= max_traversal_intersections);
  return position;
}

}  // namespace oracle
