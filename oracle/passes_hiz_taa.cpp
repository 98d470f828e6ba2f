// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// passes_hiz_taa.cpp — CPU restatement of
//   src/shaders/advanced_ssr/downsample_gbuffer.frag, depth_mips.frag  (Hi-Z build)
//   src/shaders/taa/resolve.comp                                       (TAA resolve)
#include "shader_common.hpp"

using namespace oracle;

// downsample_gbuffer.frag:12-37 (+ downsample_pass.cpp:25-92: size checks).
// Depth is compared and copied as the stored 24-bit integer: min() of already
// quantised D24 values re-quantises to itself.
extern "C" int vkr_ref_downsample_gbuffer(const vkr_img* depth, const vkr_img* normal, const vkr_img* velocity,
                                          const vkr_img* out_normal, const vkr_img* out_velocity) {
  Image d(*depth), n(*normal), v(*velocity), on(*out_normal), ov(*out_velocity);
  if (d.mips() < 2) return 1;  // "Can't downsample depth texture with 1 mip level"
  if (d.w(1) != on.w() || d.h(1) != on.h() || ov.w() != on.w() || ov.h() != on.h()) return 2;  // "Output textures have different sizes"
  const int w2 = on.w(), h2 = on.h();
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h2; y++) {
    for (int x = 0; x < w2; x++) {
      int px = 2 * x, py = 2 * y;
      auto fetch_d = [&](int lx, int ly) -> uint32_t {
        // texelFetch out of bounds -> 0 (only possible for odd extents)
        if (lx >= d.w(0) || ly >= d.h(0)) return 0u;
        return d.load_u32(lx, ly, 0) & 0xFFFFFFu;
      };
      uint32_t d0 = fetch_d(px, py), d1 = fetch_d(px + 1, py), d2 = fetch_d(px, py + 1), d3 = fetch_d(px + 1, py + 1);
      uint32_t m01 = d0 < d1 ? d0 : d1, m23 = d2 < d3 ? d2 : d3;
      uint32_t min_depth = m01 < m23 ? m01 : m23;
      int ox = 0, oy = 0;
      if (min_depth == d1) { ox = 1; oy = 0; }
      else if (min_depth == d2) { ox = 0; oy = 1; }
      else if (min_depth == d3) { ox = 1; oy = 1; }
      int sx = px + ox, sy = py + oy;
      uint32_t nt = 0, vt = 0;
      if (sx < n.w(0) && sy < n.h(0)) { nt = n.load_u32(sx, sy, 0); vt = v.load_u32(sx, sy, 0); }
      on.store_u32(x, y, 0, nt);   // RG16_UNORM -> RG16_UNORM: bit copy
      ov.store_u32(x, y, 0, vt);   // RG16F -> RG16F: bit copy
      d.store_u32(x, y, 1, min_depth);
    }
  }
  return 0;
}

// depth_mips.frag:7-15 + downsample_pass.cpp:94-131: mip i = 2x2 min of mip i-1 for
// i = src_mip+1 .. mips-1, extent max(1, W>>i); odd extents drop the last row/column,
// a 1-wide parent makes the +1 fetch fall out of bounds -> 0.
extern "C" int vkr_ref_depth_mips(const vkr_img* depth, uint32_t src_mip) {
  Image d(*depth);
  for (int i = (int)src_mip + 1; i < d.mips(); i++) {
    const int w = d.w(i), h = d.h(i), pw = d.w(i - 1), ph = d.h(i - 1);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
      for (int x = 0; x < w; x++) {
        auto fetch_d = [&](int lx, int ly) -> uint32_t {
          if (lx >= pw || ly >= ph) return 0u;
          return d.load_u32(lx, ly, i - 1) & 0xFFFFFFu;
        };
        uint32_t d0 = fetch_d(2 * x, 2 * y), d1 = fetch_d(2 * x + 1, 2 * y);
        uint32_t d2 = fetch_d(2 * x, 2 * y + 1), d3 = fetch_d(2 * x + 1, 2 * y + 1);
        uint32_t m01 = d0 < d1 ? d0 : d1, m23 = d2 < d3 ? d2 : d3;
        d.store_u32(x, y, i, m01 < m23 ? m01 : m23);
      }
    }
  }
  return 0;
}

namespace {
// resolve.comp:72-77
vec3 reconstruct_world_pos(const Image& depth_tex, const mat4& inverse_camera, vec2 screen_uv, const float* fazz) {
  float d = depth_tex.sample(screen_uv, 0).x;
  vec3 v_camera = reconstruct_view_vec(screen_uv, d, fazz[0], fazz[1], fazz[2], fazz[3]);
  vec4 v_world = inverse_camera * vec4(v_camera, 1.0f);
  return v_world.xyz();
}
}  // namespace

// resolve.comp:20-70
extern "C" int vkr_ref_taa_resolve(const vkr_img* history_color, const vkr_img* history_depth,
                                   const vkr_img* current_depth, const vkr_img* velocity, const vkr_img* color,
                                   const vkr_img* out_color, const vkr_reproject_params* params) {
  Image hist(*history_color), hd(*history_depth), cd(*current_depth), vel(*velocity), col(*color), out(*out_color);
  mat4 inv_cam, prev_inv_cam;
  std::memcpy(inv_cam.m, params->inverse_camera.m, 64);
  std::memcpy(prev_inv_cam.m, params->prev_inverse_camera.m, 64);
  const float* fazz = params->fovy_aspect_znear_zfar;
  const int tw = out.fw(), th = out.fh();
#pragma omp parallel for schedule(static)
  for (int ly = 0; ly < out.h(); ly++) {
    int gy = out.oy() + ly;
    for (int lx = 0; lx < out.w(); lx++) {
      int gx = out.ox() + lx;
      vec2 screen_uv(((float)gx + 0.5f) / (float)tw, ((float)gy + 0.5f) / (float)th);
      vec3 out_c(0, 0, 0);
      vec3 current_color = col.sample(screen_uv).xyz();
      vec2 v = vel.sample(screen_uv).xy();
      float delta_len = length(v);
      bool reprojected = false;
      vec2 prev_uv = screen_uv + v;
      if (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f) {
        vec3 history = hist.sample(prev_uv).xyz();
        vec3 color0 = hist.sample(prev_uv, 0, ivec2(1, 0)).xyz();
        vec3 color1 = hist.sample(prev_uv, 0, ivec2(0, 1)).xyz();
        vec3 color2 = hist.sample(prev_uv, 0, ivec2(-1, 0)).xyz();
        vec3 color3 = hist.sample(prev_uv, 0, ivec2(0, -1)).xyz();
        vec3 color_min = min(color0, min(color1, min(color2, color3)));
        vec3 color_max = max(color0, max(color1, max(color2, color3)));
        history = clamp(history, color_min, color_max);
        out_c = mix(history, current_color, 0.1f);
        vec3 v_world_cur = reconstruct_world_pos(cd, inv_cam, screen_uv, fazz);
        vec3 v_world_prev = reconstruct_world_pos(hd, prev_inv_cam, prev_uv, fazz);
        vec3 v_camera = (inv_cam * vec4(0, 0, 0, 1)).xyz();
        const float MAX_REPROJECTION_EPS = 0.2f, MIN_REPROJECTION_EPS = 0.01f;
        float error = length(v_world_cur - v_world_prev);
        float pixel_dist = length(v_world_cur - v_camera);
        reprojected = (delta_len < 0.005f) ||
                      (error < clamp((0.1f * pixel_dist) * delta_len, MIN_REPROJECTION_EPS, MAX_REPROJECTION_EPS));
      }
      if (!reprojected) out_c = current_color;
      out.store(gx, gy, vec4(out_c, 0.0f));
    }
  }
  return 0;
}
