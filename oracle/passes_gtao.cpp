// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// passes_gtao.cpp — CPU restatement of src/shaders/gtao/{main,filter,accum}.comp.
#include "shader_common.hpp"

using namespace oracle;

namespace {

// main.comp:276-278
inline float gtao_direction(ivec2 pos) {
  return (1.0f / 16.0f) * (float)((((pos.x + pos.y) & 3) << 2) + (pos.x & 3));
}

struct GtaoCtx {
  const Image& depth;
  const Image& gbuffer_normal;
  const Image& gbuffer_material;
  const Image& pdf;
  const Image& gtao_out;
  const vkr_gtao_params& p;
  const vkr_gtao_push& pc;
  mat4 normal_mat;
};

// main.comp:84-108
float find_horizon(const GtaoCtx& c, vec2 start, vec3 camera_start, vec2 dir, int samples_count, vec3 v) {
  const float MAX_THIKNESS = 0.1f;  // main.comp:82
  float h_cos = -1.0f;
  float previous_z = camera_start.z;
  for (int i = 1; i <= samples_count; i++) {
    vec2 tc = madd(start, (float)i / (float)samples_count, dir);
    float sample_depth = c.depth.sample(tc, 0).x;
    vec3 sample_pos = reconstruct_view_vec(tc, sample_depth, c.p.fovy, c.p.aspect, c.p.znear, c.p.zfar);
    if (sample_pos.z > previous_z + MAX_THIKNESS) break;
    previous_z = sample_pos.z;
    vec3 sample_offset = sample_pos - camera_start;
    float sample_cos = dot(v, normalize(sample_offset));
    if (sample_cos > h_cos) h_cos = sample_cos;
  }
  return h_cos;
}

// cos/sin of the slice angle: the angle takes 16 values per launch; evaluated with
// the host libm in fp32 (cosf/sinf), which the product precomputes into a table.
inline vec2 slice_dir(float angle) { return vec2(cosf(angle), sinf(angle)); }

// main.comp:185-217
float gtao_camera_space(const GtaoCtx& c, ivec2 pos, vec2 screen_uv, uint32_t dirs_count) {
  float frag_depth = c.depth.sample(screen_uv).x;
  if (frag_depth >= 1.0f) return 0.0f;
  vec3 camera_pos = reconstruct_view_vec(screen_uv, frag_depth, c.p.fovy, c.p.aspect, c.p.znear, c.p.zfar);
  vec3 w0 = -normalize(camera_pos);
  vec3 n_world = decode_normal(c.gbuffer_normal.sample(screen_uv).xy());
  vec3 camera_normal = normalize((c.normal_mat * vec4(n_world, 0.0f)).xyz());
  ivec2 ds = c.depth.size(0);
  float rad = min(100.0f / length(camera_pos), 16.0f);
  vec2 dir_radius(rad / (float)ds.x, rad / (float)ds.y);
  float base_angle = gtao_direction(pos) + c.pc.angle_offset;
  float sum = 0.0f;
  for (uint32_t dir_index = 0; dir_index < dirs_count; dir_index++) {
    float angle = (2.0f * PI) * (base_angle + (float)dir_index / (float)dirs_count);
    vec2 sample_direction = dir_radius * slice_dir(angle);
    vec3 sample_end_pos = reconstruct_view_vec(screen_uv + sample_direction, frag_depth, c.p.fovy, c.p.aspect, c.p.znear, c.p.zfar);
    vec3 slice_normal = normalize(cross(w0, -sample_end_pos));
    vec3 normal_projected = madd(camera_normal, -dot(camera_normal, slice_normal), slice_normal);
    vec3 X = -normalize(cross(slice_normal, w0));
    float n = PI / 2.0f - acosf(dot(normalize(normal_projected), X));
    float h_cos = find_horizon(c, screen_uv, camera_pos, sample_direction, 16, w0);
    float h = acosf(h_cos);
    h = min(n + min(h - n, PI / 2.0f), h);
    sum += (length(normal_projected) * 0.25f) * max((-cosf(2.0f * h - n) + cosf(n)) + (2.0f * h) * sinf(n), 0.0f);
  }
  return (2.0f * sum) / (float)dirs_count;
}

// main.comp:219-274
vec2 mis_gtao(const GtaoCtx& c, ivec2 pos, vec2 screen_uv) {
  float frag_depth = c.depth.sample(screen_uv).x;
  if (frag_depth >= 1.0f) return vec2(0.0f, 1.0f);
  vec3 camera_pos = reconstruct_view_vec(screen_uv, frag_depth, c.p.fovy, c.p.aspect, c.p.znear, c.p.zfar);
  vec3 w0 = -normalize(camera_pos);
  vec3 n_world = decode_normal(c.gbuffer_normal.sample(screen_uv).xy());
  vec3 camera_normal = normalize((c.normal_mat * vec4(n_world, 0.0f)).xyz());
  ivec2 ds = c.depth.size(0);
  float rad = min(100.0f / length(camera_pos), 16.0f);
  vec2 dir_radius(rad / (float)ds.x, rad / (float)ds.y);
  float base_angle = gtao_direction(pos) + c.pc.angle_offset;
  float angle = (2.0f * PI) * base_angle;
  vec2 sample_direction = dir_radius * slice_dir(angle);
  vec3 sample_end_pos = reconstruct_view_vec(screen_uv + sample_direction, frag_depth, c.p.fovy, c.p.aspect, c.p.znear, c.p.zfar);
  vec3 L = normalize(sample_end_pos - camera_pos);
  vec3 slice_normal = normalize(cross(w0, -sample_end_pos));
  vec3 normal_projected = madd(camera_normal, -dot(camera_normal, slice_normal), slice_normal);
  vec3 X = -normalize(cross(slice_normal, w0));
  float n = PI / 2.0f - acosf(dot(normalize(normal_projected), X));
  float h_cos = find_horizon(c, screen_uv, camera_pos, sample_direction, 16, w0);
  float h = acosf(h_cos);
  h = min(n + min(h - n, PI / 2.0f), h);
  float occlusion = (((1.0f / PI) * length(normal_projected)) * 0.25f) *
                    max((-cosf(2.0f * h - n) + cosf(n)) + (2.0f * h) * sinf(n), 0.0f);
  float roughness = c.gbuffer_material.sample(screen_uv).y;
  float pdf_ggx = sampleGGXdirPDF(c.pdf, w0, camera_normal, L, roughness * roughness);
  float pdf_uniform = 1.0f / (2.0f * PI);
  vec2 ao = c.gtao_out.fetch(pos, 0).xy();  // imageLoad
  if (c.pc.reflections_only != 0) {
    float res = ao.x / ao.y;
    return vec2(isnan(res) ? 1.0f : res, 1.0f);
  }
  float alpha = 1.0f / (c.pc.weight_ratio + 1.0f);
  float betta = 1.0f - alpha;
  float mis_weight1 = alpha / (alpha * ao.y + betta * pdf_uniform);
  float mis_weight2 = betta / (alpha * pdf_ggx + betta * pdf_uniform);
  float mis_ao = ao.x * mis_weight1 + occlusion * mis_weight2;
  float result = mis_ao;
  float total_weight = 1.0f;
  return vec2(isnan(result) ? occlusion / pdf_uniform : mis_ao, total_weight);
}

}  // namespace

// main.comp:52-67.  tex_size derives from the floor-divided dispatch (gtao.cpp:145).
extern "C" int vkr_ref_gtao_main(const vkr_img* depth, const vkr_gtao_params* params, const vkr_img* normal,
                                 const vkr_img* material, const vkr_img* pdf_tex, const vkr_img* gtao_inout,
                                 const vkr_gtao_push* push) {
  Image d(*depth), nrm(*normal), mat(*material), pdf(*pdf_tex), out(*gtao_inout);
  GtaoCtx c{d, nrm, mat, pdf, out, *params, *push, mat4()};
  std::memcpy(c.normal_mat.m, params->normal_mat.m, sizeof(float) * 16);
  const int tw = (out.fw() / 8) * 8, th = (out.fh() / 4) * 4;
  const int x0 = out.ox(), y0 = out.oy();
#pragma omp parallel for schedule(dynamic, 4)
  for (int ly = 0; ly < out.h(); ly++) {
    int gy = y0 + ly;
    if (gy >= th) continue;
    for (int lx = 0; lx < out.w(); lx++) {
      int gx = x0 + lx;
      if (gx >= tw) continue;
      UbPixel ub(UB_GTAO_MAIN);
      ivec2 pixel_pos(gx, gy);
      vec2 screen_uv(((float)gx + 0.5f) / (float)tw, ((float)gy + 0.5f) / (float)th);
      vec2 occlusion(0.0f, 1.0f / (2.0f * PI));
      if (push->use_mis > 0)
        occlusion = mis_gtao(c, pixel_pos, screen_uv);
      else
        occlusion.x = gtao_camera_space(c, pixel_pos, screen_uv, (push->two_directions != 0) ? 2u : 1u);
      out.store(gx, gy, vec4(occlusion.x, occlusion.y, 0.0f, 0.0f));
    }
  }
  return 0;
}

// gtao/filter.comp:17-51.  The sky branch stores but does not return (filter.comp:23-25),
// so its value is overwritten below — restated literally as "no early out".
extern "C" int vkr_ref_gtao_filter(const vkr_img* depth, const vkr_img* raw_gtao, const vkr_img* out_filtered,
                                   const vkr_gtao_filter_push* push) {
  Image d(*depth), raw(*raw_gtao), out(*out_filtered);
  const float znear = push->znear, zfar = push->zfar;
  const int tw = (out.fw() / 8) * 8, th = (out.fh() / 4) * 4;
#pragma omp parallel for schedule(static)
  for (int ly = 0; ly < out.h(); ly++) {
    int gy = out.oy() + ly;
    if (gy >= th) continue;
    for (int lx = 0; lx < out.w(); lx++) {
      int gx = out.ox() + lx;
      if (gx >= tw) continue;
      UbPixel ub(UB_GTAO_FILTER);
      float pixel_depth = d.fetch(gx, gy, 0).x;
      float linear_depth = linearize_depth2(pixel_depth, znear, zfar);
      float weight_sum = 0.0f, ao = 0.0f;
      for (int x = 0; x < 4; x++) {
        for (int y = 0; y < 4; y++) {
          int sx = gx + (x - 2), sy = gy + (y - 2);
          float sampled_depth = linearize_depth2(d.fetch(sx, sy, 0).x, znear, zfar);
          float weight = max(0.0f, 1.0f - (5.0f * abs(sampled_depth - linear_depth)) / abs(linear_depth));
          weight_sum += weight;
          ao += weight * raw.fetch(sx, sy, 0).x;
        }
      }
      ao /= weight_sum;
      out.store(gx, gy, vec4(ao, 0, 0, 0));
    }
  }
  return 0;
}

namespace {
// accum.comp:90-95
vec3 reconstruct_world_pos(const Image& depth_tex, const mat4& inverse_camera, vec2 screen_uv, const float* fazz, int lod) {
  float d = depth_tex.sample(screen_uv, lod).x;
  vec3 v_camera = reconstruct_view_vec(screen_uv, d, fazz[0], fazz[1], fazz[2], fazz[3]);
  vec4 v_world = inverse_camera * vec4(v_camera, 1.0f);
  return v_world.xyz();
}
}  // namespace

// gtao/accum.comp:29-88
extern "C" int vkr_ref_gtao_accumulate(const vkr_img* depth, const vkr_img* prev_depth, const vkr_img* current_ao,
                                       const vkr_img* out_accumulated, const vkr_img* velocity, const vkr_img* history,
                                       const vkr_gtao_accum_params* params, const vkr_gtao_accum_push* push) {
  Image cd(*depth), pd(*prev_depth), cur(*current_ao), out(*out_accumulated), vel(*velocity), hist(*history);
  mat4 inv_cam, prev_inv_cam, mvp;
  std::memcpy(inv_cam.m, params->inverse_camera.m, 64);
  std::memcpy(prev_inv_cam.m, params->prev_inverse_camera.m, 64);
  std::memcpy(mvp.m, params->mvp.m, 64);
  const float* fazz = params->fovy_aspect_znear_zfar;
  const float MAX_SAMPLES = 255.0f;
  const int tw = out.fw(), th = out.fh();
#pragma omp parallel for schedule(static)
  for (int ly = 0; ly < out.h(); ly++) {
    int gy = out.oy() + ly;
    for (int lx = 0; lx < out.w(); lx++) {
      int gx = out.ox() + lx;
      vec2 tex_size((float)tw, (float)th);
      vec2 screen_uv(((float)gx + 0.5f) / tex_size.x, ((float)gy + 0.5f) / tex_size.y);
      vec2 velocity_v = vel.sample(screen_uv).xy();
      vec2 prev_uv = screen_uv + velocity_v;
      bool reprojected = false;
      float valid_samples = 1.0f;
      if (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f) {
        vec3 v_world_prev = reconstruct_world_pos(pd, prev_inv_cam, prev_uv, fazz, 0);
        vec4 prev_ndc = mvp * vec4(v_world_prev, 1.0f);
        prev_ndc = prev_ndc / prev_ndc.w;
        vec2 prev_world_uv(0.5f * prev_ndc.x + 0.5f, 0.5f * prev_ndc.y + 0.5f);
        vec2 delta = abs(prev_world_uv - screen_uv) * tex_size;
        const float znear = fazz[2], zfar = fazz[3];
        float current_z = linearize_depth2(cd.sample(screen_uv).x, znear, zfar);
        float prev_z = linearize_depth2(prev_ndc.z, znear, zfar);
        float depth_err = abs(prev_z - current_z);
        float vel_delta = max(abs(velocity_v.x) * tex_size.x, abs(velocity_v.y) * tex_size.y);
        float error = 0.1f * vel_delta + depth_err;
        valid_samples = clamp(1.0f - error, 0.8f, 1.0f);
        reprojected = (max(delta.x, delta.y) <= 2.0f) && (depth_err < 0.2f);
      }
      float new_ao = cur.fetch(gx, gy, 0).x;
      float computed_ao = new_ao;
      float samples_count = 1.0f;
      if (push->clear_history != 0) reprojected = false;
      if (reprojected) {
        vec2 accumulated = hist.sample(prev_uv).xy();
        samples_count = (255.0f * accumulated.y) * valid_samples;
        computed_ao = (accumulated.x * samples_count + new_ao) / (samples_count + 1.0f);
        samples_count += 1.0f;
        if (samples_count > MAX_SAMPLES) samples_count = 100.0f;
      }
      out.store(gx, gy, vec4(clamp(computed_ao, 0.0f, 1.0f), samples_count / 255.0f, 0.0f, 0.0f));
    }
  }
  return 0;
}
