// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// passes_ssr.cpp — CPU restatement of src/shaders/advanced_ssr/
//   {preintegrate,trace,filter,blur}.comp and the host-side Halton table
//   (src/advanced_ssr.cpp:8-34).
#include "shader_common.hpp"

using namespace oracle;

// advanced_ssr.cpp:8-34: Halton(2,3) of index iter+1, with the float-floor division
// quirk (`current = floor(current / float(base))`).  out: count x vec4 (zw = 0).
extern "C" void vkr_ref_halton23(float* out_vec4, uint32_t count) {
  auto halton_elem = [](uint32_t index, uint32_t base) {
    float f = 1.0f, r = 0.0f;
    uint32_t current = index;
    do {
      f = f / (float)base;
      r = r + f * (float)(current % base);
      current = (uint32_t)floorf((float)current / (float)base);
    } while (current > 0);
    return r;
  };
  for (uint32_t i = 0; i < count; i++) {
    out_vec4[4 * i + 0] = halton_elem(i + 1, 2);
    out_vec4[4 * i + 1] = halton_elem(i + 1, 3);
    out_vec4[4 * i + 2] = 0.0f;
    out_vec4[4 * i + 3] = 0.0f;
  }
}

// preintegrate.comp:45-65 (live #else main) with G2 of :77-84
extern "C" int vkr_ref_pdf_preintegrate(const vkr_img* out_pdf) {
  Image out(*out_pdf);
  const int STEP_COUNT = 2000;
  const int rw = out.fw(), rh = out.fh();
#pragma omp parallel for schedule(static)
  for (int y = 0; y < out.h(); y++) {
    for (int x = 0; x < out.w(); x++) {
      const float a = (2.0f * ((float)x + 0.5f)) / (float)rw - 1.0f;
      const float b = ((float)y + 0.5f) / (float)rh;
      float sum = 0.0f;
      float dt = 2.0f / (float)STEP_COUNT;
      for (int i = 0; i < STEP_COUNT; i++) {
        float t = -1.0f + dt * ((float)i + 0.5f);
        const float p = b - a, q = b + a;
        const float L = p * t + q;
        const float nom = (1.0f - t) * L;
        const float denom = (1.0f + t * t) - (0.5f * L) * L;
        sum += (L > 0.0f) ? nom / (denom * denom) : 0.0f;
      }
      float g = (2.0f / (float)STEP_COUNT) * sum;
      out.store(x, y, vec4(g, 0, 0, 0));
    }
  }
  return 0;
}

namespace {

// trace.comp:143-154 (== main.comp:69-80)
vec3 get_tangent(vec3 n) {
  float max_xy = max(abs(n.x), abs(n.y));
  vec3 t;
  if (max_xy < 0.00001f) t = vec3(1, 0, 0);
  else t = vec3(n.y, -n.x, 0);
  return normalize(t);
}

// trace.comp:156-158.  sin() of a large argument decides the Halton index, so it is
// frozen as "evaluate in double, round once to float".
float rand_co(vec2 co) {
  float d = dot(co, vec2(12.9898f, 78.233f));
  float s = (float)std::sin((double)d);
  return fract(s * 43758.5453f);
}

struct TraceCtx {
  const Image& DEPTH;
  const vkr_trace_params& p;
};

// instrumentation for tools/trace_sim.py (lane-utilisation studies of the product kernel's schedule): when a sink is
// set, the march stores the number of steps every ray took (row pitch = sink_pitch bytes, window coordinates)
uint8_t* g_step_sink = nullptr;
int g_step_sink_pitch = 0;
thread_local uint32_t t_last_steps = 0;
// instrumentation for kernel design (tools/trace_steps.py --gate): how often the horizon update of a step is evaluated, passes its
// |v| < 0.3 gate, and could have been refused by |v.z| alone; [0..3] pinned steps, [4..7] later steps: {steps, mip <= 1, passed, |v.z| >= 0.3}
uint64_t* g_gate_counters = nullptr;
// instrumentation for the multi-GPU design (tools/trace_row_reach.py): per ray, the first and one-past-the-last row of pyramid
// level 0 covered by the texels its march fetched inside the frame (a texel of level L in row ty covers rows [ty << L, (ty + 1) << L));
// two uint16 per ray: {first, end}, 0xFFFF / 0 when it fetched nothing inside the frame
uint16_t* g_reach_sink = nullptr;
int g_reach_sink_pitch = 0;
thread_local uint32_t t_reach_lo = 0xFFFFu, t_reach_hi = 0u;

// trace.comp:206-268
vec3 hierarchical_raymarch_find_hor(const TraceCtx& c, vec3 origin, vec3 direction, int most_detailed_mip,
                                    uint32_t max_traversal_intersections, bool& valid_hit, vec3 w0,
                                    vec3 camera_start, float& h) {
  const Image& depth_tex = c.DEPTH;
  MarchSetup s = march_setup(depth_tex, direction, most_detailed_mip);
  int current_mip = most_detailed_mip;
  vec2 current_mip_resolution = s.res;
  vec2 current_mip_resolution_inv = s.res_inv;
  float current_t;
  vec3 position;
  initial_advance_ray(origin, direction, s.inv_direction, current_mip_resolution, current_mip_resolution_inv,
                      s.floor_offset, s.uv_offset, position, current_t);
  h = 0.0f;
  uint32_t i = 0;
  t_reach_lo = 0xFFFFu; t_reach_hi = 0u;
  while (i < max_traversal_intersections && current_mip >= most_detailed_mip) {
    vec2 current_mip_position = current_mip_resolution * position.xy();
    float surface_z = depth_tex.fetch(to_ivec2(current_mip_position), current_mip).x;
    if (g_reach_sink) {
      const ivec2 q = to_ivec2(current_mip_position);
      if (current_mip >= 0 && current_mip < depth_tex.mips() && q.x >= 0 && q.y >= 0 && q.x < depth_tex.fw(current_mip) && q.y < depth_tex.fh(current_mip)) {
        t_reach_lo = std::min(t_reach_lo, (uint32_t)q.y << current_mip);
        t_reach_hi = std::max(t_reach_hi, std::min((uint32_t)(q.y + 1) << current_mip, 0xFFFFu));
      }
    }
    bool skipped_tile = advance_ray(origin, direction, s.inv_direction, current_mip_position,
                                    current_mip_resolution_inv, s.floor_offset, s.uv_offset, surface_z,
                                    position, current_t);
    bool mip0sample = i < 15;
    current_mip += mip0sample ? 0 : (skipped_tile ? 1 : -1);
    current_mip_resolution *= mip0sample ? 1.0f : (skipped_tile ? 0.5f : 2.0f);
    current_mip_resolution_inv *= mip0sample ? 1.0f : (skipped_tile ? 2.0f : 0.5f);
    ++i;
    vec3 v = reconstruct_view_vec(position.xy(), surface_z, c.p.fovy, c.p.aspect, c.p.znear, c.p.zfar) - camera_start;
    if (current_mip <= 1) {
      float h2 = dot(w0, normalize(v));
      if (length(v) < 0.3f) h = max(h, h2);
    }
    if (g_gate_counters) {
      uint64_t* k = g_gate_counters + (i <= 15 ? 0 : 4);
      __atomic_fetch_add(&k[0], 1, __ATOMIC_RELAXED);
      if (current_mip <= 1) {
        __atomic_fetch_add(&k[1], 1, __ATOMIC_RELAXED);
        if (length(v) < 0.3f) __atomic_fetch_add(&k[2], 1, __ATOMIC_RELAXED);
        if (std::fabs(v.z) >= 0.3001f) __atomic_fetch_add(&k[3], 1, __ATOMIC_RELAXED);
      }
    }
  }
  valid_hit = (i <= max_traversal_intersections);
  t_last_steps = i;
  return position;
}

}  // namespace

extern "C" void vkr_ref_set_step_sink(uint8_t* sink, int pitch_bytes) { g_step_sink = sink; g_step_sink_pitch = pitch_bytes; }
extern "C" void vkr_ref_set_gate_counters(uint64_t* counters8) { g_gate_counters = counters8; }
extern "C" void vkr_ref_set_reach_sink(uint16_t* sink, int pitch_bytes) { g_reach_sink = sink; g_reach_sink_pitch = pitch_bytes; }

// trace.comp:41-141.  window != NULL: the multi-GPU variant (include/vkr_postfx.h vkr_sssr_trace_windowed) — the same
// conjunction of validity tests, with the hit-normal test of a ray whose footprint rows are not all inside
// [normal_row0, normal_row1) deferred to vkr_ref_sssr_validate.
static int trace_impl(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                      const vkr_trace_params* params, const float* halton_vec4,
                      const vkr_img* out_ray, const vkr_img* out_occlusion, const vkr_img* pdf_tex,
                      const vkr_trace_push* push, const vkr_trace_window_push* window, const vkr_img* pending_mask,
                      const vkr_img* pending_data) {
  Image DEPTH(*depth), NORMAL(*normal), MATERIAL(*material), OUT_RAY(*out_ray), OUT_OCC(*out_occlusion), PDF_TEX(*pdf_tex);
  const vkr_trace_params& p = *params;
  mat4 normal_mat;
  std::memcpy(normal_mat.m, p.normal_mat.m, 64);
  TraceCtx c{DEPTH, p};
  const int tw = OUT_RAY.fw(), th = OUT_RAY.fh();
#pragma omp parallel for schedule(dynamic, 2)
  for (int ly = 0; ly < OUT_RAY.h(); ly++) {
    int gy = OUT_RAY.oy() + ly;
    for (int lx = 0; lx < OUT_RAY.w(); lx++) {
      int gx = OUT_RAY.ox() + lx;
      UbPixel ub(UB_SSSR_TRACE);
      vec2 tex_size((float)tw, (float)th);
      vec2 screen_uv(((float)gx + 0.5f) / tex_size.x, ((float)gy + 0.5f) / tex_size.y);
      vec3 material_v = MATERIAL.sample(screen_uv).xyz();
      float roughness = material_v.y;
      material_v.y = mix(0.0f, push->max_roughness, roughness);
      roughness = material_v.y * material_v.y;

      float pixel_depth = DEPTH.sample(screen_uv).x;
      vec3 pixel_normal_world = sample_gbuffer_normal(NORMAL, screen_uv);
      vec3 pixel_normal = normalize((normal_mat * vec4(pixel_normal_world, 0.0f)).xyz());
      vec3 view_vec = reconstruct_view_vec(screen_uv, pixel_depth, p.fovy, p.aspect, p.znear, p.zfar);

      const uint32_t base_index = f2u(rand_co(screen_uv) * (float)VKR_HALTON_SEQ_SIZE);
      uint32_t index = (base_index + p.frame_random) & (VKR_HALTON_SEQ_SIZE - 1);
      vec2 rnd(halton_vec4[4 * index + 0], halton_vec4[4 * index + 1]);

      vec3 tangent = get_tangent(pixel_normal);
      vec3 bitangent = normalize(cross(pixel_normal, tangent));
      tangent = normalize(cross(bitangent, pixel_normal));

      vec3 view_dir = -normalize(view_vec);
      view_dir = vec3(dot(view_dir, tangent), dot(view_dir, bitangent), dot(view_dir, pixel_normal));

      vec3 brdf_norm = sampleGGXVNDF(view_dir, roughness, roughness, rnd.x, rnd.y);
      vec3 N = (brdf_norm.x * tangent + brdf_norm.y * bitangent) + brdf_norm.z * pixel_normal;
      vec3 R = reflect(view_vec, N);

      vec3 ray_start = project_view_vec(view_vec + 0.001f * pixel_normal, p.fovy, p.aspect, p.znear, p.zfar);
      ray_start.z -= 0.0001f;

      vec3 ray_dir = project_view_vec(view_vec + R, p.fovy, p.aspect, p.znear, p.zfar);
      ray_dir -= ray_start;
      ray_dir *= (1.0f - ray_start.z) / ray_dir.z;

      bool valid_hit = false;
      float h = -1.0f;
      vec3 w0 = -normalize(view_vec);

      vec3 out_r = hierarchical_raymarch_find_hor(c, ray_start, ray_dir, 0, 80, valid_hit, pixel_normal, view_vec, h);
      if (g_step_sink) g_step_sink[(size_t)ly * g_step_sink_pitch + lx] = (uint8_t)t_last_steps;
      if (g_reach_sink) {
        uint16_t* r = (uint16_t*)((uint8_t*)g_reach_sink + (size_t)ly * g_reach_sink_pitch) + 2 * lx;
        r[0] = (uint16_t)t_reach_lo; r[1] = (uint16_t)t_reach_hi;
      }

      if (valid_hit) {
        vec2 ray_step = abs(out_r.xy() - ray_start.xy()) * tex_size;
        if (max(ray_step.x, ray_step.y) < 2.0f) valid_hit = false;
      }
      auto depth_test = [&]() {
        float hit_depth = DEPTH.sample(out_r.xy(), 0).x;
        float hit_z = linearize_depth2(hit_depth, p.znear, p.zfar);
        float ray_z = linearize_depth2(out_r.z, p.znear, p.zfar);
        return !(ray_z > hit_z + 0.3f || ray_z < hit_z - 0.1f);
      };
      auto hit_normal_faces_ray = [&]() {
        vec3 hit_normal_world = sample_gbuffer_normal(NORMAL, out_r.xy());
        vec3 hit_normal = (normal_mat * vec4(hit_normal_world, 0.0f)).xyz();
        return !(dot(hit_normal, R) > 0.0f);
      };
      if (!window) {
        if (valid_hit && (!hit_normal_faces_ray() || dot(pixel_normal, R) < 0.0f)) valid_hit = false;
        if (valid_hit && !depth_test()) valid_hit = false;
      } else {
        if (valid_hit && dot(pixel_normal, R) < 0.0f) valid_hit = false;
        if (valid_hit && !depth_test()) valid_hit = false;
        bool pending = false;
        if (valid_hit) {
          const int nfh = NORMAL.fh();
          const int y0 = f2i(floorf(cfma(out_r.y, (float)nfh, -0.5f)));
          const int r0 = clamp(y0, 0, nfh - 1), r1 = clamp(y0 + 1, 0, nfh - 1);
          pending = r0 < (int)window->normal_row0 || r1 >= (int)window->normal_row1;
          if (!pending) {
            if (!hit_normal_faces_ray()) valid_hit = false;
          } else {
            float* pd = (float*)Image(*pending_data).texel_ptr(2 * lx, ly, 0);
            pd[0] = R.x; pd[1] = R.y; pd[2] = R.z; pd[3] = 0.0f;
            pd[4] = out_r.x; pd[5] = out_r.y; pd[6] = 0.0f; pd[7] = 0.0f;
          }
        }
        *Image(*pending_mask).texel_ptr(lx, ly, 0) = pending ? 1 : 0;
      }
      OUT_RAY.store(gx, gy, vec4(out_r, valid_hit ? pixel_depth : 1.0f));

      {
        vec3 slice_normal = normalize(cross(w0, R));
        vec3 normal_projected = pixel_normal - dot(pixel_normal, slice_normal) * slice_normal;
        vec3 X = normalize(cross(slice_normal, w0));
        float n = PI / 2.0f - acosf(dot(normalize(normal_projected), X));
        bool no_occlusion = h == -1.0f;
        h = acosf(h);
        h = min(n + min(h - n, PI / 2.0f), h);
        float pdf = sampleGGXdirPDF(PDF_TEX, w0, pixel_normal, R, roughness);
        float occlusion = (((1.0f / PI) * length(normal_projected)) * 0.25f) *
                          max((-cosf(2.0f * h - n) + cosf(n)) + (2.0f * h) * sinf(n), 0.0f);
        float result = isnan(occlusion) ? 0.0f : occlusion;
        OUT_OCC.store(gx, gy, vec4(no_occlusion ? 0.0f : result, no_occlusion ? 0.0f : pdf, 0, 0));
      }
    }
  }
  return 0;
}

extern "C" int vkr_ref_sssr_trace(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                                  const vkr_trace_params* params, const float* halton_vec4,
                                  const vkr_img* out_ray, const vkr_img* out_occlusion, const vkr_img* pdf_tex,
                                  const vkr_trace_push* push) {
  return trace_impl(depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, push, nullptr, nullptr, nullptr);
}

extern "C" int vkr_ref_sssr_trace_windowed(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                                           const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_ray,
                                           const vkr_img* out_occlusion, const vkr_img* pdf_tex, const vkr_img* pending_mask,
                                           const vkr_img* pending_data, const vkr_trace_window_push* push) {
  const vkr_trace_push base {push->max_roughness};
  return trace_impl(depth, normal, material, params, halton_vec4, out_ray, out_occlusion, pdf_tex, &base, push, pending_mask, pending_data);
}

// the deferred hit-normal test (trace.comp:103-109) of the pending rays
extern "C" int vkr_ref_sssr_validate(const vkr_img* rays, const vkr_img* pending_mask, const vkr_img* pending_data, const vkr_img* frame_normals,
                                     const vkr_trace_params* params) {
  Image RAYS(*rays), MASK(*pending_mask), DATA(*pending_data), NORMAL(*frame_normals);
  mat4 normal_mat;
  std::memcpy(normal_mat.m, params->normal_mat.m, 64);
  for (int ly = 0; ly < RAYS.h(); ly++)
    for (int lx = 0; lx < RAYS.w(); lx++) {
      if (*MASK.texel_ptr(lx, ly, 0) == 0) continue;
      const float* pd = (const float*)DATA.texel_ptr(2 * lx, ly, 0);
      vec3 hit_normal_world = sample_gbuffer_normal(NORMAL, vec2(pd[4], pd[5]));
      vec3 hit_normal = (normal_mat * vec4(hit_normal_world, 0.0f)).xyz();
      if (dot(hit_normal, vec3(pd[0], pd[1], pd[2])) > 0.0f) {
        uint16_t one = 0xFFFFu;
        std::memcpy(RAYS.texel_ptr(lx, ly, 0) + 6, &one, 2);
      }
    }
  return 0;
}

namespace {
// filter.comp:97-108; note the argument order of brdfG1(NdotV, alpha2) at :106 is
// swapped w.r.t. its declaration brdf.glsl:43 — restated literally.
vec3 ray_weight(vec3 N, vec3 V, vec3 L, vec3 F0, float roughness) {
  vec3 H = normalize(V + L);
  vec3 F = fresnelSchlick(max(dot(H, V), 0.0f), F0);
  float alpha2 = roughness * roughness;
  float NdotL = max(dot(N, L), 0.0f);
  float NdotV = max(dot(N, V), 0.0f);
  float G2 = brdfG2(NdotL, NdotV, alpha2);
  float G1 = brdfG1(NdotV, alpha2);
  return (F * G2) / G1;
}
}  // namespace

// filter.comp:36-91 + process_pixel :110-149 (FULL_RES 0)
extern "C" int vkr_ref_sssr_filter(const vkr_img* rays, const vkr_img* depth, const vkr_img* albedo,
                                   const vkr_img* normal, const vkr_img* material, const vkr_img* out_reflections,
                                   const vkr_trace_params* params, const vkr_filter_push* push) {
  Image RAYS(*rays), DEPTH(*depth), ALBEDO(*albedo), NORMAL(*normal), MATERIAL(*material), OUT(*out_reflections);
  const vkr_trace_params& p = *params;
  mat4 normal_mat;
  std::memcpy(normal_mat.m, p.normal_mat.m, 64);
  const uint32_t render_flags = push->render_flags;
  const int tw = OUT.fw(), th = OUT.fh();
  static const int offsets[5][2] = {{0, 0}, {-1, 0}, {0, 1}, {1, 0}, {0, -1}};
#pragma omp parallel for schedule(static)
  for (int ly = 0; ly < OUT.h(); ly++) {
    int gy = OUT.oy() + ly;
    for (int lx = 0; lx < OUT.w(); lx++) {
      int gx = OUT.ox() + lx;
      UbPixel ub(UB_SSSR_FILTER);
      vec2 tex_size((float)tw, (float)th);
      vec2 screen_uv((float)gx / tex_size.x, (float)gy / tex_size.y);
      vec4 material_v = MATERIAL.sample(screen_uv);
      const float metallic = material_v.z, roughness = material_v.y;
      vec3 albedo_v = ALBEDO.sample(screen_uv).xyz();
      vec3 F0 = F0_approximation(albedo_v, metallic);
      vec3 color_sum(0.0f), weight_sum(0.0f);
      float center_depth = DEPTH.fetch(gx, gy, 1).x;

      auto process_pixel = [&](int px, int py) {
        vec4 trace_result = RAYS.fetch(px, py, 0);
        vec2 pixel_uv((float)px / tex_size.x, (float)py / tex_size.y);
        float pixel_depth = DEPTH.fetch(px, py, 1).x;
        vec3 view_vec = reconstruct_view_vec(pixel_uv, pixel_depth, p.fovy, p.aspect, p.znear, p.zfar);
        vec3 pixel_normal = sample_gbuffer_normal(NORMAL, pixel_uv);
        pixel_normal = (normal_mat * vec4(pixel_normal, 0.0f)).xyz();
        vec3 hit_vec = reconstruct_view_vec(trace_result.xy(), trace_result.z, p.fovy, p.aspect, p.znear, p.zfar);
        vec3 radiance = (trace_result.w != 1.0f) ? ALBEDO.sample(trace_result.xy()).xyz() : vec3(0.0f);
        vec3 V = -normalize(view_vec);
        vec3 N = pixel_normal;
        vec3 L = normalize(hit_vec - view_vec);
        vec3 weight = ray_weight(N, V, L, F0, roughness);
        float bilateral_weight = 1.0f;
        if ((render_flags & VKR_BILATERAL_FILTER) != 0)
          bilateral_weight = max(1.0f - (1000.0f * abs(center_depth - pixel_depth)) / center_depth, 0.0f);
        weight *= bilateral_weight;
        color_sum += weight * radiance;
        weight_sum += weight;
      };

      if ((render_flags & VKR_NORMALIZE_REFLECTIONS) != 0) {
        for (int i = 0; i < 5; i++) process_pixel(gx + offsets[i][0], gy + offsets[i][1]);
      } else {
        process_pixel(gx, gy);
      }
      if (max(weight_sum.x, max(weight_sum.y, weight_sum.z)) < 0.001f) weight_sum = vec3(1, 1, 1);
      color_sum /= weight_sum;
      OUT.store(gx, gy, vec4(color_sum, 0.0f));
    }
  }
  return 0;
}

// blur.comp:31-115
extern "C" int vkr_ref_sssr_blur(const vkr_img* depth, const vkr_img* normal, const vkr_img* reflections,
                                 const vkr_img* material, const vkr_img* history, const vkr_img* velocity,
                                 const vkr_img* history_depth, const vkr_img* out_blurred,
                                 const vkr_reproject_params* params, const vkr_blur_push* push) {
  Image DEPTH(*depth), NORMAL(*normal), REFL(*reflections), MATERIAL(*material), HISTORY(*history), VELOCITY(*velocity),
      HDEPTH(*history_depth), OUT(*out_blurred);
  mat4 inv_cam, prev_inv_cam;
  std::memcpy(inv_cam.m, params->inverse_camera.m, 64);
  std::memcpy(prev_inv_cam.m, params->prev_inverse_camera.m, 64);
  const float* fazz = params->fovy_aspect_znear_zfar;
  const int tw = OUT.fw(), th = OUT.fh();
  // blur.comp:110-115: textureLod(depth_tex, uv, 1.0)
  auto reconstruct_world_pos = [&](const Image& depth_tex, const mat4& inverse_camera, vec2 screen_uv) {
    float d = depth_tex.sample(screen_uv, 1).x;
    vec3 v_camera = reconstruct_view_vec(screen_uv, d, fazz[0], fazz[1], fazz[2], fazz[3]);
    return (inverse_camera * vec4(v_camera, 1.0f)).xyz();
  };
#pragma omp parallel for schedule(dynamic, 2)
  for (int ly = 0; ly < OUT.h(); ly++) {
    int gy = OUT.oy() + ly;
    for (int lx = 0; lx < OUT.w(); lx++) {
      int gx = OUT.ox() + lx;
      UbPixel ub(UB_SSSR_BLUR);
      vec2 tex_size((float)tw, (float)th);
      vec2 screen_uv(((float)gx + 0.5f) / tex_size.x, ((float)gy + 0.5f) / tex_size.y);
      float roughness = MATERIAL.sample(screen_uv).y;
      roughness = mix(0.0f, push->max_roughness, roughness);
      float center_depth = DEPTH.fetch(gx, gy, 1).x;
      vec3 center_normal = sample_gbuffer_normal(NORMAL, screen_uv);
      float sigma = mix(0.4f, 4.0f, roughness);
      if (push->disable_blur != 0) sigma = 0.35f;
      float weight_sum = 0.0f;
      vec3 color(0, 0, 0);
      int r = f2i(floorf(3.0f * sigma - 0.01f));
      float g = 1.0f / (((2.0f * PI) * sigma) * sigma);
      float e = (2.0f * sigma) * sigma;
      for (int i = -r; i <= r; i++) {
        for (int j = -r; j <= r; j++) {
          int px = gx + i, py = gy + j;
          vec2 uv((float)px / tex_size.x, (float)py / tex_size.y);
          float pixel_depth = DEPTH.fetch(px, py, 1).x;
          vec3 pixel_normal = sample_gbuffer_normal(NORMAL, uv);
          float bilateral_weight = max(1.0f - (1000.0f * abs(center_depth - pixel_depth)) / center_depth, 0.0f);
          float normal_weight = max(dot(center_normal, pixel_normal), 0.0f);
          float w = g * expf((float)(-(i * i + j * j)) / e);
          w *= bilateral_weight;
          w *= normal_weight;
          color += REFL.fetch(px, py, 0).xyz() * w;
          weight_sum += w;
        }
      }
      color /= max(weight_sum, 0.001f);

      bool reprojected = false;
      vec2 v = VELOCITY.sample(screen_uv).xy();
      float delta_len = length(v);
      vec2 prev_uv = screen_uv + v;
      if (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f) {
        vec3 v_world_cur = reconstruct_world_pos(DEPTH, inv_cam, screen_uv);
        vec3 v_world_prev = reconstruct_world_pos(HDEPTH, prev_inv_cam, prev_uv);
        vec3 v_camera = (inv_cam * vec4(0, 0, 0, 1)).xyz();
        const float MAX_REPROJECTION_EPS = 0.1f, MIN_REPROJECTION_EPS = 0.01f;
        float error = length(v_world_cur - v_world_prev);
        float pixel_dist = length(v_world_cur - v_camera);
        float velocity_len = length(v);
        reprojected = (velocity_len < 0.0001f) ||
                      (error < clamp((0.1f * pixel_dist) * delta_len, MIN_REPROJECTION_EPS, MAX_REPROJECTION_EPS));
      }
      if (push->accumulate == 0) reprojected = false;
      if (reprojected) {
        vec3 history_color = HISTORY.sample(screen_uv).xyz();
        color = mix(history_color, color, 0.1f);
      }
      OUT.store(gx, gy, vec4(color, 0.0f));
    }
  }
  return 0;
}

// ---- tile-classified trace (advanced_ssr.cpp:216-302,440-495) --------------------

// advanced_ssr.cpp:447-450
extern "C" int vkr_ref_sssr_clear_indirect(uint32_t* reflective_args, uint32_t* glossy_args) {
  const uint32_t initial[3] = {0, 1, 1};
  std::memcpy(reflective_args, initial, sizeof(initial));
  std::memcpy(glossy_args, initial, sizeof(initial));
  return 0;
}

// classification.comp:38-98.  Tiles are visited in row-major order here; the shader appends them in
// whatever order its atomics resolve, so only the *set* of each list is defined.
extern "C" int vkr_ref_sssr_classification(const vkr_img* material, int32_t* reflective_tiles, int32_t* glossy_tiles,
                                           uint32_t* reflective_args, uint32_t* glossy_args,
                                           const vkr_classification_push* push) {
  Image MATERIAL_TEX(*material);
  const int TILE = 8;
  const int W = push->width, H = push->height;
  const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
  for (int ty = 0; ty < tiles_y; ty++) {
    for (int tx = 0; tx < tiles_x; tx++) {
      float g_roughness[64];
      for (int t = 0; t < 64; t++) {
        const int px = tx * TILE + (t & 7), py = ty * TILE + (t >> 3);
        float sampled_roughness = 1.0f;
        if (px < W && py < H) sampled_roughness = MATERIAL_TEX.sample(vec2((float)px / (float)W, (float)py / (float)H)).y;
        g_roughness[t] = mix(0.0f, push->max_roughness, sampled_roughness);
      }
      for (uint32_t offset = 32; offset != 0; offset /= 2)  // :72-80, all threads of a step read before any writes
        for (uint32_t t = 0; t < offset; t++) g_roughness[t] += g_roughness[t + offset];
      const float average_roughness = g_roughness[0] / 64.0f;
      const int tile_index = ty * tiles_x + tx;
      if (average_roughness < push->glossy_value) reflective_tiles[reflective_args[0]++] = tile_index;
      else glossy_tiles[glossy_args[0]++] = tile_index;
    }
  }
  return 0;
}

// trace_indirect.comp:43-135, for the indirect_args[0] tiles of the list
extern "C" int vkr_ref_sssr_trace_indirect(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                                           const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_rays,
                                           const int32_t* tiles, const uint32_t* indirect_args, uint32_t max_tiles,
                                           const vkr_trace_indirect_push* push) {
  Image DEPTH(*depth), NORMAL(*normal), MATERIAL(*material), OUT_RAY(*out_rays);
  const vkr_trace_params& p = *params;
  mat4 normal_mat;
  std::memcpy(normal_mat.m, p.normal_mat.m, 64);
  const int TILE = 8;
  const int tw = OUT_RAY.fw(), th = OUT_RAY.fh();
  const int tile_width = (tw + TILE - 1) / TILE;
  const uint32_t count = indirect_args[0] < max_tiles ? indirect_args[0] : max_tiles;
#pragma omp parallel for schedule(dynamic, 4)
  for (uint32_t g = 0; g < count; g++) {
    const int tile_index = tiles[g];
    for (int t = 0; t < TILE * TILE; t++) {
      const int gx = TILE * (tile_index % tile_width) + (t & 7), gy = TILE * (tile_index / tile_width) + (t >> 3);
      const vec2 tex_size((float)tw, (float)th);
      const vec2 screen_uv((float)gx / tex_size.x, (float)gy / tex_size.y);
      if (gx >= tw || gy >= th) continue;
      float roughness = MATERIAL.sample(screen_uv).y;
      roughness = mix(0.0f, push->max_roughness, roughness);
      roughness *= roughness;
      const float pixel_depth = DEPTH.sample(screen_uv).x;
      const vec3 pixel_normal_world = sample_gbuffer_normal(NORMAL, screen_uv);
      const vec3 pixel_normal = normalize((normal_mat * vec4(pixel_normal_world, 0.0f)).xyz());
      const vec3 view_vec = reconstruct_view_vec(screen_uv, pixel_depth, p.fovy, p.aspect, p.znear, p.zfar);
      const uint32_t base_index = f2u(rand_co(screen_uv) * (float)VKR_HALTON_SEQ_SIZE);
      const uint32_t index = (base_index + p.frame_random) & (VKR_HALTON_SEQ_SIZE - 1);
      const vec2 rnd(halton_vec4[4 * index + 0], halton_vec4[4 * index + 1]);
      vec3 tangent = get_tangent(pixel_normal);
      const vec3 bitangent = normalize(cross(pixel_normal, tangent));
      tangent = normalize(cross(bitangent, pixel_normal));
      vec3 view_dir = -normalize(view_vec);
      view_dir = vec3(dot(view_dir, tangent), dot(view_dir, bitangent), dot(view_dir, pixel_normal));
      const vec3 brdf_norm = sampleGGXVNDF(view_dir, roughness, roughness, rnd.x, rnd.y);
      const vec3 N = (brdf_norm.x * tangent + brdf_norm.y * bitangent) + brdf_norm.z * pixel_normal;
      const vec3 R = reflect(view_vec, N);
      vec3 ray_start = project_view_vec(view_vec + 0.001f * pixel_normal, p.fovy, p.aspect, p.znear, p.zfar);
      ray_start.z -= 0.0001f;
      vec3 ray_dir = project_view_vec(view_vec + R, p.fovy, p.aspect, p.znear, p.zfar);
      ray_dir -= ray_start;
      ray_dir *= (1.0f - ray_start.z) / ray_dir.z;
      bool valid_hit = false;
      vec3 out_r;
      if (push->reflection_type == 0) out_r = hierarchical_raymarch(DEPTH, ray_start, ray_dir, 0, 50, valid_hit);
      else out_r = hierarchical_raymarch(DEPTH, ray_start, ray_dir, 1, 25, valid_hit);
      if (valid_hit) {
        const vec2 ray_step = abs(out_r.xy() - ray_start.xy()) * tex_size;
        if (max(ray_step.x, ray_step.y) < 2.0f) valid_hit = false;
      }
      if (valid_hit) {
        const vec3 hit_normal_world = sample_gbuffer_normal(NORMAL, out_r.xy());
        const vec3 hit_normal = (normal_mat * vec4(hit_normal_world, 0.0f)).xyz();
        if (dot(hit_normal, R) > 0.0f || dot(pixel_normal, R) < 0.0f) valid_hit = false;
      }
      if (valid_hit && push->reflection_type == 0) {
        const float hit_depth = DEPTH.sample(out_r.xy(), 0).x;
        const float hit_z = linearize_depth2(hit_depth, p.znear, p.zfar);
        const float ray_z = linearize_depth2(out_r.z, p.znear, p.zfar);
        if (ray_z > hit_z + 0.3f || ray_z < hit_z - 0.1f) valid_hit = false;
      }
      OUT_RAY.store(gx, gy, vec4(out_r, valid_hit ? pixel_depth : 1.0f));
    }
  }
  return 0;
}
