// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// passes_secondary.cpp — CPU restatement of the passes the reference ships but does not call from
// its frame loop: the simple mirror SSR (src/ssr.cpp + shaders/ssr/shader.frag).
#include "shader_common.hpp"

using namespace oracle;

namespace {
// texture() through the depth sampler of ssr.cpp:21-28: NEAREST, U/W clamp-to-border (opaque black
// -> depth 0), V clamp-to-edge.
float sample_depth_nearest(const Image& depth, vec2 uv, int mip) {
  const int w = depth.fw(mip), h = depth.fh(mip);
  const int x = f2i(floorf(uv.x * (float)w));
  int y = f2i(floorf(uv.y * (float)h));
  if (x < 0 || x >= w) return 0.0f;
  y = clamp(y, 0, h - 1);
  return depth.fetch(x, y, mip).x;
}
// GLSL smoothstep
float smoothstep1(float e0, float e1, float x) {
  float t = clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return (t * t) * (3.0f - 2.0f * t);
}
}  // namespace

// ssr/shader.frag:27-102.  screen_uv of the full-screen triangle at pixel (x,y) is frozen as
// ((x + 0.5)/W, (y + 0.5)/H).
extern "C" int vkr_ref_ssr(const vkr_img* normal, const vkr_img* depth, const vkr_img* frame, const vkr_ssr_params* params,
                           const vkr_img* material, const vkr_img* out) {
  Image normal_tex(*normal), depth_tex(*depth), frame_tex(*frame), material_tex(*material), OUT(*out);
  mat4 camera_normal;
  std::memcpy(camera_normal.m, params->normal_mat.m, 64);
  const float fovy = params->fovy, aspect = params->aspect, znear = params->znear, zfar = params->zfar;
  const int ow = OUT.fw(), oh = OUT.fh();
#pragma omp parallel for schedule(dynamic, 2)
  for (int ly = 0; ly < OUT.h(); ly++) {
    const int gy = OUT.oy() + ly;
    for (int lx = 0; lx < OUT.w(); lx++) {
      const int gx = OUT.ox() + lx;
      vec4 out_reflection(0, 0, 0, 0);
      do {
        const vec2 screen_uv(((float)gx + 0.5f) / (float)ow, ((float)gy + 0.5f) / (float)oh);
        const ivec2 fs = frame_tex.size(0);
        const vec2 tex_size((float)fs.x, (float)fs.y);
        const vec2 aligned_screen_uv = floor(screen_uv * tex_size) / tex_size + vec2(0.5f / tex_size.x, 0.5f / tex_size.y);
        const vec3 material_v = material_tex.sample(screen_uv).xyz();
        const float roughness = material_v.y;
        const float pixel_depth = sample_depth_nearest(depth_tex, aligned_screen_uv, 0);
        const vec3 pixel_normal_world = sample_gbuffer_normal(normal_tex, aligned_screen_uv);
        const vec3 pixel_normal = normalize((camera_normal * vec4(pixel_normal_world, 0.0f)).xyz());
        const vec3 view_vec = reconstruct_view_vec(aligned_screen_uv, pixel_depth, fovy, aspect, znear, zfar);
        const vec3 R = reflect(view_vec, pixel_normal);
        const vec3 H = pixel_normal;
        const vec3 start = project_view_vec(view_vec + 0.0005f * pixel_normal, fovy, aspect, znear, zfar);
        const vec3 p = project_view_vec(view_vec + R, fovy, aspect, znear, zfar);
        vec3 delta = normalize(p - start);
        if (abs(delta.z) < 0.0000001f) break;
        float t_bound = (1.0f - start.z) / delta.z;
        const float u_bound = max((1.0f - start.x) / delta.x, -start.x / delta.x);
        const float v_bound = max((1.0f - start.y) / delta.y, -start.y / delta.y);
        t_bound = min(t_bound, min(u_bound, v_bound));
        const vec3 end = start + t_bound * delta;
        bool valid_hit = false;
        const vec3 out_ray = hierarchical_raymarch(depth_tex, start, end - start, 0, 100, valid_hit);
        if (!valid_hit) break;
        const vec2 screen_size = tex_size;
        const vec2 dist0 = abs(out_ray.xy() - start.xy());
        const vec2 min_dist(2.0f / screen_size.x, 2.0f / screen_size.y);
        if (dist0.x < min_dist.x && dist0.y < min_dist.y) break;
        const vec3 hit_normal_world = sample_gbuffer_normal(normal_tex, out_ray.xy());
        const vec3 hit_normal = (camera_normal * vec4(hit_normal_world, 0.0f)).xyz();
        if (dot(hit_normal, R) > 0.0f) break;
        const float hit_depth = sample_depth_nearest(depth_tex, out_ray.xy(), 0);
        if (out_ray.z > hit_depth + 0.0001f) break;
        const vec2 fov(0.05f * (screen_size.y / screen_size.x), 0.05f * 1.0f);
        const float bx = smoothstep1(0.0f, fov.x, out_ray.x) * (1.0f - smoothstep1(1.0f - fov.x, 1.0f, out_ray.x));
        const float by = smoothstep1(0.0f, fov.y, out_ray.y) * (1.0f - smoothstep1(1.0f - fov.y, 1.0f, out_ray.y));
        const float coef = bx * by;
        const vec4 c = frame_tex.sample(out_ray.xy());
        const float k = DistributionGGX(pixel_normal, H, roughness);
        const float ndr = max(dot(pixel_normal, R), 0.0f);
        out_reflection = ((coef * c) * k) * ndr;
      } while (false);
      OUT.store(gx, gy, out_reflection);
    }
  }
  return 0;
}
