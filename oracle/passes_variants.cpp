// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED.
//
// passes_variants.cpp — CPU restatement of the passes the reference ships but its frame loop never
// records:
//   gtao/main.frag, gtao/reproject.comp, gtao_opt/{deinterleave,main_deinterleaved}.comp,
//   screen_trace/{trace,filter,accumulate}.comp.
// They are restated literally, including their dispatch quirks (see each function).
#include <vector>

#include "shader_common.hpp"

using namespace oracle;

namespace {

// main.frag:188-190 (== main.comp:276-278)
inline float gtao_direction(ivec2 pos) {
  return (1.0f / 16.0f) * (float)((((pos.x + pos.y) & 3) << 2) + (pos.x & 3));
}
// the slice angle takes 16 values per launch: host libm in fp32, a table in the product
inline vec2 slice_dir(float angle) { return vec2(cosf(angle), sinf(angle)); }

struct Proj4 { float fovy, aspect, znear, zfar; };

// A "depth sampler": texture(depth, uv) and textureSize(depth, 0) for either a D24 view or one
// layer of the R32F deinterleaved array.
struct DepthSampler {
  const Image& img;
  float sample(vec2 uv) const { return img.sample(uv, 0).x; }
  ivec2 size() const { return img.size(0); }
};

// main.frag:66-90 == main_deinterleaved.comp:54-76 (MAX_THIKNESS 0.1)
float find_horizon(const DepthSampler& depth, const Proj4& p, vec2 start, vec3 camera_start, vec2 dir, int samples_count, vec3 v) {
  const float MAX_THIKNESS = 0.1f;
  float h_cos = -1.0f;
  float previous_z = camera_start.z;
  for (int i = 1; i <= samples_count; i++) {
    vec2 tc = start + ((float)i / (float)samples_count) * dir;
    float sample_depth = depth.sample(tc);
    vec3 sample_pos = reconstruct_view_vec(tc, sample_depth, p.fovy, p.aspect, p.znear, p.zfar);
    if (sample_pos.z > previous_z + MAX_THIKNESS) break;
    previous_z = sample_pos.z;
    vec3 sample_offset = sample_pos - camera_start;
    float sample_cos = dot(v, normalize(sample_offset));
    if (sample_cos > h_cos) h_cos = sample_cos;
  }
  return h_cos;
}

// main.frag:164-196 == main_deinterleaved.comp:86-116: one slice set, 20 samples, radius
// min(200/|P|, 32) texels of the sampled depth, sky -> 1.
float gtao_camera_space_v2(const DepthSampler& depth, const Image& gbuffer_normal, const mat4& normal_mat, const Proj4& p,
                           float angle_offset, ivec2 pos, vec2 screen_uv, uint32_t dirs_count) {
  const int SAMPLES = 20;
  float frag_depth = depth.sample(screen_uv);
  if (frag_depth >= 1.0f) return 1.0f;
  vec3 camera_pos = reconstruct_view_vec(screen_uv, frag_depth, p.fovy, p.aspect, p.znear, p.zfar);
  vec3 w0 = -normalize(camera_pos);
  vec3 camera_normal = normalize((normal_mat * vec4(decode_normal(gbuffer_normal.sample(screen_uv).xy()), 0.0f)).xyz());
  ivec2 ds = depth.size();
  float rad = min(200.0f / length(camera_pos), 32.0f);
  vec2 dir_radius(rad / (float)ds.x, rad / (float)ds.y);
  float base_angle = gtao_direction(pos) + angle_offset;
  float sum = 0.0f;
  for (uint32_t dir_index = 0; dir_index < dirs_count; dir_index++) {
    float angle = (2.0f * PI) * (base_angle + (float)dir_index / (float)dirs_count);
    vec2 sample_direction = dir_radius * slice_dir(angle);
    vec3 sample_end_pos = reconstruct_view_vec(screen_uv + sample_direction, frag_depth, p.fovy, p.aspect, p.znear, p.zfar);
    vec3 slice_normal = normalize(cross(w0, -sample_end_pos));
    vec3 normal_projected = camera_normal - dot(camera_normal, slice_normal) * slice_normal;
    float n = PI / 2.0f - acosf(dot(normalize(normal_projected), normalize(sample_end_pos - camera_pos)));
    float h_cos = find_horizon(depth, p, screen_uv, camera_pos, sample_direction, SAMPLES, w0);
    float h = acosf(h_cos);
    h = min(n + min(h - n, PI / 2.0f), h);
    sum += (length(normal_projected) * 0.25f) * max((-cosf(2.0f * h - n) + cosf(n)) + (2.0f * h) * sinf(n), 0.0f);
  }
  return (2.0f * sum) / (float)dirs_count;
}

mat4 load_mat(const vkr_mat4& m) {
  mat4 r;
  std::memcpy(r.m, m.m, 64);
  return r;
}

}  // namespace

// gtao/main.frag:45-48.  Fragment (x, y) of the full-screen triangle has screen_uv frozen as
// ((x + 0.5)/W, (y + 0.5)/H); the colour attachment is RGBA16F and the shader writes one float:
// the unwritten components are frozen to 0.
extern "C" int vkr_ref_gtao_main_graphics(const vkr_img* depth, const vkr_gtao_params* params, const vkr_img* normal,
                                          const vkr_img* out_raw, const vkr_gtao_gfx_push* push) {
  Image d(*depth), nrm(*normal), out(*out_raw);
  const DepthSampler ds{d};
  const mat4 normal_mat = load_mat(params->normal_mat);
  const Proj4 p{params->fovy, params->aspect, params->znear, params->zfar};
  const int w = out.fw(), h = out.fh();
#pragma omp parallel for schedule(dynamic, 4)
  for (int ly = 0; ly < out.h(); ly++) {
    const int gy = out.oy() + ly;
    for (int lx = 0; lx < out.w(); lx++) {
      const int gx = out.ox() + lx;
      const vec2 screen_uv(((float)gx + 0.5f) / (float)w, ((float)gy + 0.5f) / (float)h);
      const float occlusion = gtao_camera_space_v2(ds, nrm, normal_mat, p, push->angle_offset, ivec2(gx, gy), screen_uv, 1);
      out.store(gx, gy, vec4(occlusion, 0.0f, 0.0f, 0.0f));
    }
  }
  return 0;
}

// gtao/reproject.comp:27-66, REPROJECT_MODE == STATIC_REPROJECT (:6).  Floor dispatch gtao.cpp:282.
extern "C" int vkr_ref_gtao_reproject(const vkr_gtao_reprojection* params, const vkr_img* depth, const vkr_img* prev_depth,
                                      const vkr_img* current_ao, const vkr_img* prev_ao, const vkr_img* out_img) {
  Image cd(*depth), pd(*prev_depth), cur(*current_ao), prev(*prev_ao), out(*out_img);
  const float REPROJECT_BIAS = 1e-6f, REPROJECT_COEF = 0.05f;
  const float fovy = params->fovy, aspect = params->aspect, znear = params->znear, zfar = params->zfar;
  const int tw = (out.fw() / 8) * 8, th = (out.fh() / 4) * 4;
#pragma omp parallel for schedule(static)
  for (int gy = 0; gy < th; gy++) {
    for (int gx = 0; gx < tw; gx++) {
      const vec2 screen_uv((float)gx / (float)tw, (float)gy / (float)th);
      const float new_ao = cur.fetch(gx, gy, 0).x;
      const float current_depth = cd.fetch(gx, gy, 0).x;
      const vec3 cur_view = reconstruct_view_vec(screen_uv, current_depth, fovy, aspect, znear, zfar);
      float ao = new_ao;
      const float sampled_depth = pd.fetch(gx, gy, 0).x;
      const float sampled_ao = prev.fetch(gx, gy, 0).x;
      const float sampled_z = linearize_depth2(sampled_depth, znear, zfar);
      const float delta = abs(sampled_z - cur_view.z);
      if (delta < REPROJECT_BIAS && sampled_depth < 1.0f) ao = mix(sampled_ao, new_ao, REPROJECT_COEF);
      out.store(gx, gy, vec4(ao, 0, 0, 0));
    }
  }
  return 0;
}

// gtao_opt/deinterleave.comp:10-21.  Dispatch (layer_w/8, layer_h/4) as recorded at gtao.cpp:468:
// pixel_pos spans the *layer* extent, not the depth extent.
extern "C" int vkr_ref_deinterleave_depth(const vkr_img* depth, const vkr_img* layers, uint32_t layer_count,
                                          const vkr_deinterleave_push* push) {
  if (layer_count == 0) return 1;
  Image d(*depth);
  const int step = push->pattern_step;
  const int tw = (int)(layers[0].full_width / 8) * 8, th = (int)(layers[0].full_height / 4) * 4;
  const int pattern_mod = (1 << step) - 1;
  for (int y = 0; y < th; y++) {
    for (int x = 0; x < tw; x++) {
      const float sampled_depth = d.fetch(x, y, 0).x;
      const int ox = x >> step, oy = y >> step;
      const int layer = ((y & pattern_mod) << step) + (x & pattern_mod);
      if (layer < 0 || layer >= (int)layer_count) continue;
      Image L(layers[layer]);
      if (ox >= L.fw() || oy >= L.fh()) continue;
      L.store(ox, oy, vec4(sampled_depth, 0, 0, 0));
    }
  }
  return 0;
}

// gtao_opt/main_deinterleaved.comp:38-52.  Dispatch (out_w/8, out_h/4) (gtao.cpp:517-521, one
// dispatch per array layer *of the output image*); imageStore outside `out` is dropped.
extern "C" int vkr_ref_gtao_main_deinterleaved(const vkr_img* layers, uint32_t layer_count, const vkr_gtao_params* params,
                                               const vkr_img* normal, const vkr_img* out_raw,
                                               const vkr_gtao_deinterleaved_push* push) {
  if (layer_count == 0) return 1;
  const uint32_t layer = push->layer < layer_count ? push->layer : layer_count - 1;  // array layer clamps
  Image L(layers[layer]), nrm(*normal), out(*out_raw);
  const DepthSampler ds{L};
  const mat4 normal_mat = load_mat(params->normal_mat);
  const Proj4 p{params->fovy, params->aspect, params->znear, params->zfar};
  const int scale = 1 << push->pattern_n, scale_mod = scale - 1;
  const int gw = (out.fw() / 8) * 8, gh = (out.fh() / 4) * 4;  // invocations
  const int tw = scale * gw, th = scale * gh;
#pragma omp parallel for schedule(dynamic, 4)
  for (int iy = 0; iy < gh; iy++) {
    for (int ix = 0; ix < gw; ix++) {
      const ivec2 pixel_pos(scale * ix + (int)(push->layer & (uint32_t)scale_mod),
                            scale * iy + (int)((push->layer >> push->pattern_n) & (uint32_t)scale_mod));
      if (pixel_pos.x >= out.fw() || pixel_pos.y >= out.fh()) continue;  // dropped store; no other side effect
      const vec2 screen_uv((float)pixel_pos.x / (float)tw, (float)pixel_pos.y / (float)th);
      const float occlusion = gtao_camera_space_v2(ds, nrm, normal_mat, p, push->angle_offset, pixel_pos, screen_uv, 1);
      out.store(pixel_pos.x, pixel_pos.y, vec4(occlusion, 0, 0, 0));
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// ScreenSpaceTrace
namespace {

const int TILE_SIZE = 8;

// trace.comp:45-47; sin() frozen as in passes_ssr.cpp: evaluated in double, rounded once
float rand_co(vec2 co) {
  float d = dot(co, vec2(12.9898f, 78.233f));
  float s = (float)std::sin((double)d);
  return fract(s * 43758.5453f);
}

// trace.comp:213-226
void calc_tangent_space(vec3 normal, vec3& tangent, vec3& bitangent) {
  if (abs(normal.z) > 0.0f) {
    float k = sqrtf(normal.y * normal.y + normal.z * normal.z);
    tangent = vec3(0.0f, -normal.z / k, normal.y / k);
  } else {
    float k = sqrtf(normal.x * normal.x + normal.y * normal.y);
    tangent = vec3(normal.y / k, -normal.x / k, 0.0f);
  }
  bitangent = cross(normal, tangent);
}

struct TileSlot {
  bool alive = false;  // false: the invocation returned at trace.comp:232-235 (sky)
  vec3 hit_pos = vec3(-1, -1, -1);
  vec3 hit_color = vec3(0, 0, 0);
  vec3 camera_pos, camera_normal;
  vec2 uv;
  float a = 0.0f;
};

}  // namespace

// screen_trace/trace.comp:27-37 + trace_tangent_space :230-343, dirs_count = 1.
extern "C" int vkr_ref_screen_trace_main(const vkr_img* depth, const vkr_img* normal, const vkr_img* color,
                                         const vkr_img* material, const vkr_img* out_raw, const vkr_screen_trace_params* params) {
  Image DEPTH(*depth), NORMAL(*normal), COLOR(*color), MATERIAL(*material), OUT(*out_raw);
  const mat4 normal_mat = load_mat(params->normal_mat);
  const float fovy = params->fovy, aspect = params->aspect, znear = params->znear, zfar = params->zfar;
  const float MAX_THIKNESS = 0.2f;
  const int SAMPLES = 20, FAR_SAMPLES = 8;
  const int groups_x = OUT.fw() / TILE_SIZE, groups_y = OUT.fh() / TILE_SIZE;
  const int tw = groups_x * TILE_SIZE, th = groups_y * TILE_SIZE;
  auto sample_normal = [&](vec2 uv) {
    return normalize((normal_mat * vec4(decode_normal(NORMAL.sample(uv).xy()), 0.0f)).xyz());
  };
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int by = 0; by < groups_y; by++) {
    for (int bx = 0; bx < groups_x; bx++) {
      TileSlot slots[TILE_SIZE * TILE_SIZE];
      // phase 1: every invocation up to the shared-memory stores (trace.comp:230-309)
      for (int ty = 0; ty < TILE_SIZE; ty++) {
        for (int tx = 0; tx < TILE_SIZE; tx++) {
          TileSlot& S = slots[ty * TILE_SIZE + tx];
          const ivec2 pos(bx * TILE_SIZE + tx, by * TILE_SIZE + ty);
          const vec2 screen_uv((float)pos.x / (float)tw, (float)pos.y / (float)th);
          S.uv = screen_uv;
          vec3 screen_pos(screen_uv.x, screen_uv.y, DEPTH.sample(screen_uv, 0).x);
          if (screen_pos.z >= 1.0f) continue;
          S.alive = true;
          vec3 camera_pos = reconstruct_view_vec(screen_uv, screen_pos.z, fovy, aspect, znear, zfar);
          const vec3 camera_normal = sample_normal(screen_uv);
          camera_pos += 1e-6f * camera_normal;
          vec3 tangent, bitangent;
          calc_tangent_space(camera_normal, tangent, bitangent);
          const float base_angle = gtao_direction(pos) + params->angle_offset;
          const float normal_angle = (PI / 2.0f) * rand_co(screen_uv + vec2(params->random_offset, 0.0f));
          const float sin_normal_angle = (float)std::sin((double)normal_angle);  // steers the ray: frozen like rand_co
          const ivec2 dsz = DEPTH.size(0);
          const float rad = min(200.0f / length(camera_pos), 32.0f);
          const vec2 ao_dir_radius(rad / (float)dsz.x, rad / (float)dsz.y);

          const float angle = (2.0f * PI) * (base_angle + 0.0f / 1.0f);
          const vec2 cs = slice_dir(angle);
          const vec3 camera_sample_dir = normalize((cs.x * tangent + cs.y * bitangent) + camera_normal * sin_normal_angle);
          vec3 screen_dir = project_view_vec(camera_pos + camera_sample_dir, fovy, aspect, znear, zfar);
          screen_dir -= screen_pos;
          screen_dir = (screen_dir / length(vec2(screen_dir.x, screen_dir.y))) * max(ao_dir_radius.x, ao_dir_radius.y);
          bool ray_hit = false;
          vec3 hit_pos(0, 0, 0);
          float h_cos = 0.0f;
          float previous_z = camera_pos.z;
          for (int i = 0; i < SAMPLES; i++) {
            const vec3 tc = screen_pos + ((float)i / (float)SAMPLES) * screen_dir;
            const float tc_depth = DEPTH.sample(vec2(tc.x, tc.y), 0).x;
            const vec3 camera_sample = reconstruct_view_vec(vec2(tc.x, tc.y), tc_depth, fovy, aspect, znear, zfar);
            if (tc.x < 0.0f || tc.x > 1.0f || tc.y < 0.0f || tc.y > 1.0f || camera_sample.z > previous_z + MAX_THIKNESS) break;
            if (!ray_hit && tc.z - 1e-6f > tc_depth) {
              hit_pos = tc;
              ray_hit = true;
            }
            h_cos = max(h_cos, dot(camera_normal, normalize(camera_sample - camera_pos)));
            previous_z = camera_sample.z;
          }
          h_cos = min(h_cos, 1.0f);
          const float h = acosf(h_cos);
          S.a = 0.25f * (1.0f - cosf(2.0f * h));
          const vec3 start_ray = screen_pos + screen_dir;
          screen_dir *= 2.0f;
          for (int i = 0; i < FAR_SAMPLES; i++) {
            const vec3 tc = start_ray + ((float)i / (float)FAR_SAMPLES) * screen_dir;
            const float tc_depth = DEPTH.sample(vec2(tc.x, tc.y), 0).x;
            const float camera_z = linearize_depth2(tc_depth, znear, zfar);
            if (tc.x < 0.0f || tc.x > 1.0f || tc.y < 0.0f || tc.y > 1.0f || camera_z > previous_z + 0.1f) break;
            if (!ray_hit && tc.z - 1e-6f > tc_depth) {
              hit_pos = tc;
              ray_hit = true;
            }
            previous_z = camera_z;
          }
          const vec3 hit_normal = ray_hit ? sample_normal(vec2(hit_pos.x, hit_pos.y)) : vec3(0, 0, 0);
          ray_hit = ray_hit && (dot(camera_normal, hit_normal) < 0.0f);
          S.hit_pos = ray_hit ? hit_pos : vec3(-1, -1, -1);
          S.hit_color = ray_hit ? COLOR.sample(vec2(hit_pos.x, hit_pos.y)).xyz() : vec3(0, 0, 0);
          S.camera_pos = camera_pos;
          S.camera_normal = camera_normal;
        }
      }
      // phase 2: 3x3 neighbourhood inside the tile (trace.comp:313-337), as if barrier() separated
      // the phases; slots of returned (sky) invocations read as "no hit".
      for (int ty = 0; ty < TILE_SIZE; ty++) {
        for (int tx = 0; tx < TILE_SIZE; tx++) {
          const TileSlot& S = slots[ty * TILE_SIZE + tx];
          const int gx = bx * TILE_SIZE + tx, gy = by * TILE_SIZE + ty;
          if (!S.alive) {
            OUT.store(gx, gy, vec4(0.0f, 0.0f, 0.0f, 1.0f));
            continue;
          }
          const vec3 W0 = -normalize(S.camera_pos);
          float weight = 0.0f;
          const float roughness = MATERIAL.sample(S.uv).y;
          vec3 accum(0, 0, 0);
          for (int x = tx - 1; x <= tx + 1; x++) {
            for (int y = ty - 1; y <= ty + 1; y++) {
              if (x >= 0 && x < TILE_SIZE && y >= 0 && y < TILE_SIZE) {
                const TileSlot& Nb = slots[y * TILE_SIZE + x];
                const vec3 rh = Nb.hit_pos;
                if (rh.z >= 0.0f) {
                  const vec3 camera_hit_pos = reconstruct_view_vec(vec2(rh.x, rh.y), rh.z, fovy, aspect, znear, zfar);
                  const vec3 L = normalize(camera_hit_pos - S.camera_pos);
                  const vec3 H = normalize(W0 + L);
                  const float w = DistributionGGX(S.camera_normal, H, roughness) * max(dot(S.camera_normal, L), 0.0f);
                  weight += w;
                  accum += Nb.hit_color * w;
                }
              }
            }
          }
          vec4 result(0, 0, 0, S.a);
          if (weight > 0.0f) {
            const vec3 r = accum / weight;
            result.x = r.x; result.y = r.y; result.z = r.z;
          }
          result.w *= 2.0f / 1.0f;
          OUT.store(gx, gy, result);
        }
      }
    }
  }
  return 0;
}

// screen_trace/filter.comp:13-39.  Like gtao/filter.comp the sky branch stores without returning
// (:20-22), so it has no effect.  linear_depth is negative, so the divisor (:32) is too and every
// weight is >= 1 — restated literally.
extern "C" int vkr_ref_screen_trace_filter(const vkr_img* raw, const vkr_img* depth, const vkr_img* out_filtered,
                                           const vkr_screen_trace_filter_push* push) {
  Image RAW(*raw), DEPTH(*depth), OUT(*out_filtered);
  const float znear = push->znear, zfar = push->zfar;
  const int tw = (OUT.fw() / 8) * 8, th = (OUT.fh() / 4) * 4;
#pragma omp parallel for schedule(static)
  for (int gy = 0; gy < th; gy++) {
    for (int gx = 0; gx < tw; gx++) {
      const float pixel_depth = DEPTH.fetch(gx, gy, 0).x;
      const float linear_depth = linearize_depth2(pixel_depth, znear, zfar);
      float weight_sum = 0.0f;
      vec4 sum(0, 0, 0, 0);
      for (int x = 0; x < 4; x++) {
        for (int y = 0; y < 4; y++) {
          const int sx = gx + (x - 2), sy = gy + (y - 2);
          const float sampled_depth = linearize_depth2(DEPTH.fetch(sx, sy, 0).x, znear, zfar);
          const float weight = max(0.0f, 1.0f - abs(sampled_depth - linear_depth) / (linear_depth * 0.1f));
          weight_sum += weight;
          const vec4 t = RAW.fetch(sx, sy, 0);
          sum = vec4(sum.x + weight * t.x, sum.y + weight * t.y, sum.z + weight * t.z, sum.w + weight * t.w);
        }
      }
      sum = vec4(sum.x / weight_sum, sum.y / weight_sum, sum.z / weight_sum, sum.w / weight_sum);
      OUT.store(gx, gy, sum);
    }
  }
  return 0;
}

// screen_trace/accumulate.comp:21-40
extern "C" int vkr_ref_screen_trace_accumulate(const vkr_img* depth, const vkr_img* prev_depth, const vkr_img* current,
                                               const vkr_img* accum_inout, const vkr_screen_trace_accum_push* push) {
  Image CD(*depth), PD(*prev_depth), CUR(*current), ACC(*accum_inout);
  const float REPROJECT_BIAS = 1e-6f, REPROJECT_COEF = 0.05f;
  const float fovy = push->fovy, aspect = push->aspect, znear = push->znear, zfar = push->zfar;
  const int tw = (ACC.fw() / 8) * 8, th = (ACC.fh() / 4) * 4;
#pragma omp parallel for schedule(static)
  for (int gy = 0; gy < th; gy++) {
    for (int gx = 0; gx < tw; gx++) {
      const vec2 screen_uv((float)gx / (float)tw, (float)gy / (float)th);
      const vec4 new_sum = CUR.fetch(gx, gy, 0);
      const float current_depth = CD.fetch(gx, gy, 0).x;
      const vec3 cur_view = reconstruct_view_vec(screen_uv, current_depth, fovy, aspect, znear, zfar);
      vec4 out_sum = new_sum;
      const float sampled_depth = PD.fetch(gx, gy, 0).x;
      const vec4 sampled_sum = ACC.fetch(gx, gy, 0);
      const float sampled_z = linearize_depth2(sampled_depth, znear, zfar);
      const float delta = abs(sampled_z - cur_view.z);
      if (delta < REPROJECT_BIAS && sampled_depth < 1.0f) out_sum = mix(sampled_sum, new_sum, REPROJECT_COEF);
      ACC.store(gx, gy, out_sum);
    }
  }
  return 0;
}
