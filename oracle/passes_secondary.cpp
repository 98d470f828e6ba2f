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

// ---- deferred shading composite ------------------------------------------------
// preintegrate_ssr.comp:12-44: split-sum LUT (A, B) over 128 VNDF samples, RG16F 1024x1024.
extern "C" int vkr_ref_brdf_preintegrate(const float* halton_vec4, const vkr_img* out_brdf) {
  Image out(*out_brdf);
  const int NUM_SAMPLES = 128;
  const int tw = out.fw(), th = out.fh();
#pragma omp parallel for schedule(static)
  for (int y = 0; y < out.h(); y++) {
    for (int x = 0; x < out.w(); x++) {
      const float roughness = ((float)x + 0.5f) / (float)tw;
      const float NdotV = ((float)y + 0.5f) / (float)th;
      const float roughness2 = roughness * roughness;
      const vec3 V(sqrtf(1.0f - NdotV * NdotV), 0.0f, NdotV);
      float A = 0.0f, B = 0.0f;
      for (int i = 0; i < NUM_SAMPLES; i++) {
        const vec3 H = sampleGGXVNDF(V, roughness2, roughness2, halton_vec4[4 * i + 0], halton_vec4[4 * i + 1]);
        const vec3 L = normalize(reflect(-V, H));
        const float NdotL = L.z;
        const float alpha = powf(1.0f - dot(V, H), 5.0f);
        const float G1 = brdfG1(roughness2, NdotV);
        const float G2 = brdfG2(NdotV, NdotL, roughness2);
        A += (G2 / G1) * (1.0f - alpha);
        B += (G2 / G1) * alpha;
      }
      A *= 1.0f / (float)NUM_SAMPLES;
      B *= 1.0f / (float)NUM_SAMPLES;
      out.store(x, y, vec4(A, B, 0, 0));
    }
  }
  return 0;
}

// defered_shading/shader.frag:41-130.  screen_uv of the full-screen triangle = ((x+.5)/W, (y+.5)/H).
// shadow_map (binding 5) is bound by the reference but never sampled by the shader.
extern "C" int vkr_ref_defered_shading(const vkr_img* albedo, const vkr_img* normal, const vkr_img* material, const vkr_img* depth,
                                       const vkr_shading_params* consts, const vkr_img* occlusion, const vkr_img* brdf,
                                       const vkr_img* reflections, const vkr_img* out, const vkr_shading_push* push) {
  Image albedo_tex(*albedo), normal_tex(*normal), material_tex(*material), depth_tex(*depth), occlusion_tex(*occlusion),
      brdf_tex(*brdf), reflections_tex(*reflections), OUT(*out);
  mat4 inverse_camera;
  std::memcpy(inverse_camera.m, consts->inverse_camera.m, 64);
  const float fovy = consts->fovy, aspect = consts->aspect, znear = consts->znear, zfar = consts->zfar;
  const vec3 LIGHT_POS(-1.85867f, 5.81832f, -0.247114f);
  const vec3 LIGHT_RADIANCE(0.1f, 0.1f, 0.1f);
  const int ow = OUT.fw(), oh = OUT.fh();
#pragma omp parallel for schedule(static)
  for (int ly = 0; ly < OUT.h(); ly++) {
    const int gy = OUT.oy() + ly;
    for (int lx = 0; lx < OUT.w(); lx++) {
      const int gx = OUT.ox() + lx;
      const vec2 screen_uv(((float)gx + 0.5f) / (float)ow, ((float)gy + 0.5f) / (float)oh);
      const vec3 normal_v = sample_gbuffer_normal(normal_tex, screen_uv);
      const vec3 albedo_v = albedo_tex.sample(screen_uv).xyz();
      const vec4 material_v = material_tex.sample(screen_uv);
      const float depth_v = depth_tex.sample(screen_uv, 0).x;
      // sample_ocllusion_ssr (:103-130): nearest-depth 2x2 upsample of the half-res AO / reflections
      vec4 ssr_occlusion(0, 0, 0, 0);
      {
        const int offs[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}};
        float delta[4];
        for (int k = 0; k < 4; k++) delta[k] = abs(depth_tex.sample(screen_uv, 1, ivec2(offs[k][0], offs[k][1])).x - depth_v);
        const float min_delta = min(min(delta[0], delta[1]), min(delta[2], delta[3]));
        int pick = 3;
        if (min_delta == delta[0]) pick = 0;
        else if (min_delta == delta[1]) pick = 1;
        else if (min_delta == delta[2]) pick = 2;
        const ivec2 o(offs[pick][0], offs[pick][1]);
        ssr_occlusion.w = occlusion_tex.sample(screen_uv, 0, o).x;
        const vec3 r = reflections_tex.sample(screen_uv, 0, o).xyz();
        ssr_occlusion.x = r.x; ssr_occlusion.y = r.y; ssr_occlusion.z = r.z;
      }
      const float occlusion_v = ssr_occlusion.w;
      const vec3 reflection = ssr_occlusion.xyz();
      const vec3 camera_view_vec = reconstruct_view_vec(screen_uv, depth_v, fovy, aspect, znear, zfar);
      const vec3 world_pos = (inverse_camera * vec4(camera_view_vec, 1.0f)).xyz();
      const vec3 camera_pos = (inverse_camera * vec4(0, 0, 0, 1)).xyz();
      const float metallic = mix(0.1f, 1.0f, material_v.z);
      const float roughness = material_v.y;
      const vec3 V = normalize(camera_pos - world_pos);
      const vec3 N = normal_v;
      const vec3 F0 = F0_approximation(albedo_v, metallic);
      vec3 Lo(0.0f);
      const vec3 L = normalize(LIGHT_POS - world_pos);
      const vec3 H = normalize(V + L);
      const float light_distance = length(LIGHT_POS - world_pos);
      const vec3 radiance = LIGHT_RADIANCE * min(100.0f / (light_distance * light_distance), 100.0f);
      const float NdotL = max(dot(N, L), 0.0f);
      const float NdotV = max(dot(N, V), 0.0f);
      const float NDF = DistributionGGX(N, H, roughness);
      const float G = brdfG2(NdotV, NdotL, roughness * roughness);
      const vec3 F = fresnelSchlick(max(dot(H, V), 0.0f), F0);
      const vec3 kS = F;
      const vec3 kD = (vec3(1.0f) - kS) * (1.0f - metallic);
      const vec3 specular = ((NDF * G) * F) / ((4.0f * NdotV) * NdotL + 0.0001f);
      const float biased_rougness = mix(push->min_max_roughness[0], push->min_max_roughness[1], roughness);
      const vec2 ssr_brdf = brdf_tex.sample(vec2(biased_rougness, NdotV)).xy();
      Lo += (((kD * albedo_v) / PI + specular) * radiance) * NdotL;
      Lo += reflection * (F0 * ssr_brdf.x + vec3(ssr_brdf.y));
      const vec3 color = occlusion_v * (vec3(0.6f) * albedo_v + Lo);
      if (push->show_ao != 0) OUT.store(gx, gy, vec4(occlusion_v, occlusion_v, occlusion_v, 0.0f));
      else OUT.store(gx, gy, vec4(color, 0.0f));
    }
  }
  return 0;
}
