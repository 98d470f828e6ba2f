// variants.hip — the passes the reference ships but its frame loop never records (SURVEY.md 8(a)
// rows G4 and R2): programs "gtao_main" (graphics GTAO), "gtao_reproject", "deinterleave_depth",
// "main_deinterleaved", "screen_trace_main", "screen_trace_filter", "screen_trace_accumulate".
//
// Reference: src/gtao.cpp:241-284,349-526, shaders/gtao/{main.frag,reproject.comp},
// shaders/gtao_opt/*.comp, src/screen_trace.cpp, shaders/screen_trace/*.comp.
// These are off the measured frame chain; they follow the frozen fp32 contract of vkr_device.hpp
// (exact IEEE sequence for everything that places a sample or decides a branch, libm for the smooth
// angles) and sample straight from global memory — L1/L2 serve the 20-tap horizon walks.
#include "vkr_host.hpp"

namespace vkr {

// ---- GTAO, second formulation (main.frag / main_deinterleaved.comp) -----------------------------
struct Gtao2Args {
  Tex depth, normal, out;
  Mat4 normal_mat;
  Proj pr;
  float slice_cs[16][2];  // (cos, sin) of 2*PI*(k/16 + angle_offset), host libm
  int inv_w, inv_h;       // invocations (graphics: the framebuffer; deinterleaved: floor dispatch)
  int scale, off_x, off_y;  // pixel_pos = scale * invocation + offset
  int tex_w, tex_h;       // divisor of screen_uv
  int half_texel;         // 1: uv = (pos + 0.5) / tex (fragment), 0: uv = pos / tex (compute)
};

template <class F> VKR_DEV float find_horizon20(const Tex& depth, const Proj& pr, f2 start, f3 camera_start, f2 dir, f3 v) {
  float h_cos = -1.0f;
  float previous_z = camera_start.z;
#pragma unroll 1
  for (int i = 1; i <= 20; i++) {
    const f2 tc = start + ((float)i / 20.0f) * dir;
    const float sample_depth = sample<F>(depth, tc);
    const f3 sample_pos = reconstruct_view_vec(tc, sample_depth, pr);
    if (sample_pos.z > previous_z + 0.1f) break;  // MAX_THIKNESS, main.frag:64
    previous_z = sample_pos.z;
    const f3 sample_offset = sample_pos - camera_start;
    // max()-reduced cosine: hardware rsq (the break test above stays exact)
    h_cos = vmax(h_cos, dot(v, sample_offset) * fast_rsq(dot(sample_offset, sample_offset)));
  }
  return h_cos;
}

// main.frag:164-186 == main_deinterleaved.comp:86-116, dirs_count = 1
template <class F> __global__ __launch_bounds__(256) void k_gtao_v2(Gtao2Args a) {
  const int ix = blockIdx.x * blockDim.x + threadIdx.x;
  const int iy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ix >= a.inv_w || iy >= a.inv_h) return;
  const int gx = a.scale * ix + a.off_x, gy = a.scale * iy + a.off_y;
  if (gx >= a.out.fw || gy >= a.out.fh) return;  // imageStore outside the image is dropped
  const int lx = gx - a.out.ox, ly = gy - a.out.oy;
  if (lx < 0 || ly < 0 || lx >= a.out.w || ly >= a.out.h) return;
  const float half = a.half_texel ? 0.5f : 0.0f;
  const f2 screen_uv = mk2(((float)gx + half) / (float)a.tex_w, ((float)gy + half) / (float)a.tex_h);
  float occlusion = 1.0f;
  const float frag_depth = sample<F>(a.depth, screen_uv);
  if (frag_depth < 1.0f) {
    const f3 camera_pos = reconstruct_view_vec(screen_uv, frag_depth, a.pr);
    const f3 w0 = -normalize(camera_pos);
    const f3 n_world = decode_normal(sample<FmtRG16U>(a.normal, screen_uv));
    const f3 camera_normal = normalize(xyz(mul(a.normal_mat, mk4(n_world.x, n_world.y, n_world.z, 0.0f))));
    const float rad = vmin(200.0f / length(camera_pos), 32.0f);
    const f2 dir_radius = mk2(rad / (float)a.depth.fw, rad / (float)a.depth.fh);
    const int dir_slot = (((gx + gy) & 3) << 2) + (gx & 3);
    const f2 sample_direction = dir_radius * mk2(a.slice_cs[dir_slot][0], a.slice_cs[dir_slot][1]);
    const f3 sample_end_pos = reconstruct_view_vec(screen_uv + sample_direction, frag_depth, a.pr);
    // The argument of this acos reaches +-1 when the projected normal lines up with the slice
    // direction (main.frag:181 normalises both and takes acos of their dot): whether it lands on
    // 1.0000001 (NaN) or 0.99999994 is decided by the last bit, so the whole chain stays exact.
    const f3 slice_normal = normalize(cross(w0, -sample_end_pos));
    const f3 normal_projected = camera_normal - dot(camera_normal, slice_normal) * slice_normal;
    const float n = VKR_PI / 2.0f - acosf(dot(normalize(normal_projected), normalize(sample_end_pos - camera_pos)));
    const float h_cos = find_horizon20<F>(a.depth, a.pr, screen_uv, camera_pos, sample_direction, w0);
    float h = acosf(h_cos);
    h = vmin(n + vmin(h - n, VKR_PI / 2.0f), h);
    const float sum = (length(normal_projected) * 0.25f) * vmax((-cosf(2.0f * h - n) + cosf(n)) + (2.0f * h) * sinf(n), 0.0f);
    occlusion = 2.0f * sum;
  }
  uint2 o;
  o.x = float_to_half_bits(occlusion);  // (occlusion, 0, 0, 0)
  o.y = 0u;
  *texel_ptr<uint2>(a.out, lx, ly) = o;
}

// reproject.comp:27-66 (STATIC_REPROJECT)
__global__ __launch_bounds__(256) void k_gtao_reproject(Tex depth, Tex prev_depth, Tex cur_ao, Tex prev_ao, Tex out, Proj pr, int tex_w, int tex_h) {
  const int lx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ly = blockIdx.y * blockDim.y + threadIdx.y;
  if (lx >= out.w || ly >= out.h) return;
  const int gx = out.ox + lx, gy = out.oy + ly;
  if (gx >= tex_w || gy >= tex_h) return;
  const float new_ao = fetch<FmtR16F>(cur_ao, gx, gy);
  // cur_view.z == linearize_depth2(current_depth) (gbuffer_encode.glsl:58-69): only z is used
  const float cur_z = linearize_depth2_unorm(fetch<FmtD24>(depth, gx, gy), pr.znear, pr.zfar);
  const float sampled_depth = fetch<FmtD24>(prev_depth, gx, gy);
  const float sampled_z = linearize_depth2_unorm(sampled_depth, pr.znear, pr.zfar);
  float ao = new_ao;
  if (fabsf(sampled_z - cur_z) < 1e-6f && sampled_depth < 1.0f) ao = mixf(fetch<FmtR16F>(prev_ao, gx, gy), new_ao, 0.05f);
  *texel_ptr<uint16_t>(out, lx, ly) = (uint16_t)float_to_half_bits(ao);
}

// deinterleave.comp:10-21.  Layers are separate descriptors with one common extent / pitch.
struct LayerSet {
  uint8_t* base[64];
  int pitch, w, h, count;
};
__global__ __launch_bounds__(256) void k_deinterleave(Tex depth, LayerSet layers, int step, int tex_w, int tex_h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= tex_w || y >= tex_h) return;
  const float sampled_depth = fetch<FmtD24>(depth, x, y);
  const int mod = (1 << step) - 1;
  const int layer = ((y & mod) << step) + (x & mod);
  const int ox = x >> step, oy = y >> step;
  if (layer >= layers.count || ox >= layers.w || oy >= layers.h) return;
  *(float*)(layers.base[layer] + (size_t)oy * layers.pitch + (size_t)ox * 4) = sampled_depth;
}

// ---- ScreenSpaceTrace ---------------------------------------------------------------------------
struct ScreenTraceArgs {
  Tex depth, normal, color, material, out;
  Mat4 normal_mat;
  Proj pr;
  float slice_cs[16][2];
  float random_offset;
  int tex_w, tex_h;
};

VKR_DEV f3 st_sample_normal(const ScreenTraceArgs& a, f2 uv) {
  const f3 n = decode_normal(sample<FmtRG16U>(a.normal, uv));
  return normalize(xyz(mul(a.normal_mat, mk4(n.x, n.y, n.z, 0.0f))));
}
// brdf.glsl:31-38 with hardware rcp (a smooth weight)
VKR_DEV float ggx_d_fast(f3 N, f3 H, float alpha) {
  const float NoH = dot(N, H), alpha2 = alpha * alpha, NoH2 = NoH * NoH;
  const float den = NoH2 * alpha2 + (1.0f - NoH2);
  return (NoH2 > 0.0f ? alpha2 : 0.0f) * fast_rcp((VKR_PI * den) * den);
}

// trace.comp:27-37,230-343.  A workgroup is four 8x8 tiles (one wave each); the tile exchange of
// RAY_HIT_POS / RAY_HIT_COLOR goes through LDS with a barrier between the phases.
#define ST_TILE 8
#define ST_TILES_X 4
__global__ __launch_bounds__(ST_TILE * ST_TILE * ST_TILES_X) void k_screen_trace(ScreenTraceArgs a) {
  __shared__ float s_hit[ST_TILES_X][ST_TILE * ST_TILE][6];  // hit uv, depth, colour
  const int tile = threadIdx.x / (ST_TILE * ST_TILE);
  const int lane = threadIdx.x % (ST_TILE * ST_TILE);
  const int tx = lane % ST_TILE, ty = lane / ST_TILE;
  const int gx = (blockIdx.x * ST_TILES_X + tile) * ST_TILE + tx;
  const int gy = blockIdx.y * ST_TILE + ty;
  const bool in_grid = gx < a.tex_w && gy < a.tex_h;
  const f2 screen_uv = mk2((float)gx / (float)a.tex_w, (float)gy / (float)a.tex_h);
  bool alive = false;
  f3 camera_pos = mk3(0, 0, 0), camera_normal = mk3(0, 0, 0);
  float res_a = 0.0f;
  f3 hit_pos = mk3(-1, -1, -1), hit_color = mk3(0, 0, 0);
  if (in_grid) {
    const f3 screen_pos = mk3(screen_uv.x, screen_uv.y, sample<FmtD24>(a.depth, screen_uv));
    if (screen_pos.z < 1.0f) {
      alive = true;
      camera_pos = reconstruct_view_vec(screen_uv, screen_pos.z, a.pr);
      camera_normal = st_sample_normal(a, screen_uv);
      camera_pos = camera_pos + 1e-6f * camera_normal;
      // calc_tangent_space, trace.comp:213-226
      f3 tangent;
      if (fabsf(camera_normal.z) > 0.0f) {
        const float k = sqrtf(camera_normal.y * camera_normal.y + camera_normal.z * camera_normal.z);
        tangent = mk3(0.0f, -camera_normal.z / k, camera_normal.y / k);
      } else {
        const float k = sqrtf(camera_normal.x * camera_normal.x + camera_normal.y * camera_normal.y);
        tangent = mk3(camera_normal.y / k, -camera_normal.x / k, 0.0f);
      }
      const f3 bitangent = cross(camera_normal, tangent);
      // rand() and sin(normal_angle) steer the ray: evaluated in double and rounded once
      const float rdot = dot(mk2(screen_uv.x + a.random_offset, screen_uv.y + 0.0f), mk2(12.9898f, 78.233f));
      const float rnd01 = fractf((float)sin((double)rdot) * 43758.5453f);
      const float normal_angle = (VKR_PI / 2.0f) * rnd01;
      const float sin_na = (float)sin((double)normal_angle);
      const float rad = vmin(200.0f / length(camera_pos), 32.0f);
      const f2 ao_dir_radius = mk2(rad / (float)a.depth.fw, rad / (float)a.depth.fh);
      const int dir_slot = (((gx + gy) & 3) << 2) + (gx & 3);
      const float cs_x = a.slice_cs[dir_slot][0], cs_y = a.slice_cs[dir_slot][1];
      const f3 camera_sample_dir = normalize((cs_x * tangent + cs_y * bitangent) + camera_normal * sin_na);
      f3 screen_dir = project_view_vec(camera_pos + camera_sample_dir, a.pr) - screen_pos;
      screen_dir = (screen_dir / length(mk2(screen_dir.x, screen_dir.y))) * vmax(ao_dir_radius.x, ao_dir_radius.y);
      bool ray_hit = false;
      f3 hp = mk3(0, 0, 0);
      float h_cos = 0.0f;
      float previous_z = camera_pos.z;
#pragma unroll 1
      for (int i = 0; i < 20; i++) {
        const f3 tc = screen_pos + ((float)i / 20.0f) * screen_dir;
        const float tc_depth = sample<FmtD24>(a.depth, mk2(tc.x, tc.y));
        const f3 camera_sample = reconstruct_view_vec(mk2(tc.x, tc.y), tc_depth, a.pr);
        if (tc.x < 0.0f || tc.x > 1.0f || tc.y < 0.0f || tc.y > 1.0f || camera_sample.z > previous_z + 0.2f) break;
        if (!ray_hit && tc.z - 1e-6f > tc_depth) {
          hp = tc;
          ray_hit = true;
        }
        // exact: 1 - cos(2 acos(h_cos)) amplifies the last bit of h_cos near 1
        h_cos = vmax(h_cos, dot(camera_normal, normalize(camera_sample - camera_pos)));
        previous_z = camera_sample.z;
      }
      h_cos = vmin(h_cos, 1.0f);
      const float h = acosf(h_cos);
      res_a = 0.25f * (1.0f - cosf(2.0f * h));
      const f3 start_ray = screen_pos + screen_dir;
      screen_dir = screen_dir * 2.0f;
#pragma unroll 1
      for (int i = 0; i < 8; i++) {
        const f3 tc = start_ray + ((float)i / 8.0f) * screen_dir;
        const float tc_depth = sample<FmtD24>(a.depth, mk2(tc.x, tc.y));
        const float camera_z = linearize_depth2(tc_depth, a.pr.znear, a.pr.zfar);
        if (tc.x < 0.0f || tc.x > 1.0f || tc.y < 0.0f || tc.y > 1.0f || camera_z > previous_z + 0.1f) break;
        if (!ray_hit && tc.z - 1e-6f > tc_depth) {
          hp = tc;
          ray_hit = true;
        }
        previous_z = camera_z;
      }
      if (ray_hit) {
        const f3 hit_normal = st_sample_normal(a, mk2(hp.x, hp.y));
        ray_hit = dot(camera_normal, hit_normal) < 0.0f;
      }
      if (ray_hit) {
        hit_pos = hp;
        hit_color = sample<FmtSRGB8>(a.color, mk2(hp.x, hp.y));
      }
    }
  }
  float* slot = s_hit[tile][lane];
  slot[0] = hit_pos.x; slot[1] = hit_pos.y; slot[2] = hit_pos.z;
  slot[3] = hit_color.x; slot[4] = hit_color.y; slot[5] = hit_color.z;
  __syncthreads();
  if (!in_grid) return;
  const int lx = gx - a.out.ox, ly = gy - a.out.oy;
  if (lx < 0 || ly < 0 || lx >= a.out.w || ly >= a.out.h) return;
  f4 result = mk4(0.0f, 0.0f, 0.0f, 1.0f);
  if (alive) {
    const f3 W0 = -normalize_fast(camera_pos);
    const float roughness = sample<FmtSRGB8>(a.material, screen_uv).y;
    float weight = 0.0f;
    f3 accum = mk3(0, 0, 0);
    for (int x = tx - 1; x <= tx + 1; x++) {
      for (int y = ty - 1; y <= ty + 1; y++) {
        if (x >= 0 && x < ST_TILE && y >= 0 && y < ST_TILE) {
          const float* nb = s_hit[tile][y * ST_TILE + x];
          if (nb[2] >= 0.0f) {
            const f3 camera_hit_pos = reconstruct_view_vec(mk2(nb[0], nb[1]), nb[2], a.pr);
            const f3 L = normalize_fast(camera_hit_pos - camera_pos);
            const f3 H = normalize_fast(W0 + L);
            const float w = ggx_d_fast(camera_normal, H, roughness) * vmax(dot(camera_normal, L), 0.0f);
            weight += w;
            accum = accum + mk3(nb[3], nb[4], nb[5]) * w;
          }
        }
      }
    }
    result = mk4(0.0f, 0.0f, 0.0f, res_a * 2.0f);
    if (weight > 0.0f) {
      const float inv = fast_rcp(weight);
      result.x = accum.x * inv; result.y = accum.y * inv; result.z = accum.z * inv;
    }
  }
  uint2 o;
  o.x = float_to_half_bits(result.x) | (float_to_half_bits(result.y) << 16);
  o.y = float_to_half_bits(result.z) | (float_to_half_bits(result.w) << 16);
  *texel_ptr<uint2>(a.out, lx, ly) = o;
}

// screen_trace/filter.comp:13-39: {linear depth, raw rgba} tile with the (-2..+1) apron in LDS
#define SF_BX 64
#define SF_BY 4
#define SF_TW (SF_BX + 3)
#define SF_TH (SF_BY + 3)
__global__ __launch_bounds__(SF_BX * SF_BY) void k_screen_trace_filter(Tex raw, Tex depth, Tex out, int tex_w, int tex_h, float znear, float zfar) {
  __shared__ float s_z[SF_TW * SF_TH];
  __shared__ f4 s_raw[SF_TW * SF_TH];
  const int tid = threadIdx.y * SF_BX + threadIdx.x;
  const int bx0 = out.ox + blockIdx.x * SF_BX - 2, by0 = out.oy + blockIdx.y * SF_BY - 2;
  for (int t = tid; t < SF_TW * SF_TH; t += SF_BX * SF_BY) {
    const int px = bx0 + t % SF_TW, py = by0 + t / SF_TW;
    s_z[t] = linearize_depth2_unorm(fetch<FmtD24>(depth, px, py), znear, zfar);
    s_raw[t] = fetch<FmtRGBA16F>(raw, px, py);
  }
  __syncthreads();
  const int lx = blockIdx.x * SF_BX + threadIdx.x;
  const int ly = blockIdx.y * SF_BY + threadIdx.y;
  if (lx >= out.w || ly >= out.h) return;
  const int gx = out.ox + lx, gy = out.oy + ly;
  if (gx >= tex_w || gy >= tex_h) return;
  const int tc = (threadIdx.y + 2) * SF_TW + (threadIdx.x + 2);
  const float linear_depth = s_z[tc];
  const float divisor = linear_depth * 0.1f;
  float weight_sum = 0.0f;
  f4 sum = mk4(0, 0, 0, 0);
#pragma unroll
  for (int x = 0; x < 4; x++) {
#pragma unroll
    for (int y = 0; y < 4; y++) {
      const int t = tc + (x - 2) + (y - 2) * SF_TW;
      const float weight = vmax(0.0f, 1.0f - fabsf(s_z[t] - linear_depth) / divisor);
      weight_sum += weight;
      const f4 r = s_raw[t];
      sum = mk4(sum.x + weight * r.x, sum.y + weight * r.y, sum.z + weight * r.z, sum.w + weight * r.w);
    }
  }
  sum = sum / weight_sum;
  uint2 o;
  o.x = float_to_half_bits(sum.x) | (float_to_half_bits(sum.y) << 16);
  o.y = float_to_half_bits(sum.z) | (float_to_half_bits(sum.w) << 16);
  *texel_ptr<uint2>(out, lx, ly) = o;
}

// screen_trace/accumulate.comp:21-40 (in place)
__global__ __launch_bounds__(256) void k_screen_trace_accumulate(Tex depth, Tex prev_depth, Tex cur, Tex acc, int tex_w, int tex_h, float znear, float zfar) {
  const int lx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ly = blockIdx.y * blockDim.y + threadIdx.y;
  if (lx >= acc.w || ly >= acc.h) return;
  const int gx = acc.ox + lx, gy = acc.oy + ly;
  if (gx >= tex_w || gy >= tex_h) return;
  const f4 new_sum = fetch<FmtRGBA16F>(cur, gx, gy);
  const float cur_z = linearize_depth2_unorm(fetch<FmtD24>(depth, gx, gy), znear, zfar);
  const float sampled_depth = fetch<FmtD24>(prev_depth, gx, gy);
  const float sampled_z = linearize_depth2_unorm(sampled_depth, znear, zfar);
  f4 out_sum = new_sum;
  uint2* dst = texel_ptr<uint2>(acc, lx, ly);
  if (fabsf(sampled_z - cur_z) < 1e-6f && sampled_depth < 1.0f) {
    const uint2 p = *dst;
    const f4 sampled_sum = mk4(half_bits_to_float(p.x & 0xFFFFu), half_bits_to_float(p.x >> 16), half_bits_to_float(p.y & 0xFFFFu), half_bits_to_float(p.y >> 16));
    out_sum = mix4(sampled_sum, new_sum, 0.05f);
  }
  uint2 o;
  o.x = float_to_half_bits(out_sum.x) | (float_to_half_bits(out_sum.y) << 16);
  o.y = float_to_half_bits(out_sum.z) | (float_to_half_bits(out_sum.w) << 16);
  *dst = o;
}

static void fill_slice_table(float (*cs)[2], float angle_offset) {
  const float PI = 3.1415926535897932384626433832795f;
  for (int k = 0; k < 16; k++) {
    const float base_angle = (1.0f / 16.0f) * (float)k + angle_offset;
    const float angle = (2.0f * PI) * (base_angle + 0.0f / 1.0f);
    cs[k][0] = cosf(angle);
    cs[k][1] = sinf(angle);
  }
}
static void fill_proj(Proj& pr, float fovy, float aspect, float znear, float zfar) {
  pr.tg = tanf(fovy / 2.0f);
  pr.aspect = aspect; pr.znear = znear; pr.zfar = zfar;
}
static int make_layers(const vkr_img* layers, uint32_t count, const char* what, Tex* first, LayerSet* set) {
  if (!layers || count == 0 || count > 64) { set_error("%s: needs 1..64 array layers", what); return VKR_ERR_NULL; }
  VKR_TRY(make_tex(&layers[0], 0, VKR_FMT_R32_SFLOAT, what, first));
  set->pitch = first->pitch; set->w = first->w; set->h = first->h; set->count = (int)count;
  for (uint32_t i = 0; i < count; i++) {
    Tex t;
    VKR_TRY(make_tex(&layers[i], 0, VKR_FMT_R32_SFLOAT, what, &t));
    if (t.w != first->w || t.h != first->h || t.pitch != first->pitch || t.ox != 0 || t.oy != 0) {
      set_error("%s: array layers must share one extent and pitch", what);
      return VKR_ERR_EXTENT;
    }
    set->base[i] = const_cast<uint8_t*>(t.p);
  }
  return VKR_OK;
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_gtao_main_graphics(const vkr_img* depth, const vkr_gtao_params* params, const vkr_img* normal,
                                      const vkr_img* out_raw, const vkr_gtao_gfx_push* push, void* stream) {
  if (!params || !push) { set_error("gtao_main: NULL params"); return VKR_ERR_NULL; }
  Gtao2Args a;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "gtao_main.depth", &a.depth));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "gtao_main.normal", &a.normal));
  VKR_TRY(make_tex(out_raw, 0, VKR_FMT_RGBA16_SFLOAT, "gtao_main.out", &a.out));
  load_mat(a.normal_mat, params->normal_mat);
  fill_proj(a.pr, params->fovy, params->aspect, params->znear, params->zfar);
  fill_slice_table(a.slice_cs, push->angle_offset);
  a.inv_w = a.out.fw; a.inv_h = a.out.fh;
  a.scale = 1; a.off_x = 0; a.off_y = 0;
  a.tex_w = a.out.fw; a.tex_h = a.out.fh;
  a.half_texel = 1;
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_gtao_v2<FmtD24>, grid2d(a.inv_w, a.inv_h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("gtao_main");
}

extern "C" int vkr_gtao_reproject(const vkr_gtao_reprojection* params, const vkr_img* depth, const vkr_img* prev_depth,
                                  const vkr_img* current_ao, const vkr_img* prev_ao, const vkr_img* out_img, void* stream) {
  if (!params) { set_error("gtao_reproject: NULL params"); return VKR_ERR_NULL; }
  Tex d, pd, cur, prev, out;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "gtao_reproject.depth", &d));
  VKR_TRY(make_tex(prev_depth, 0, VKR_FMT_D24_UNORM_S8, "gtao_reproject.prev_depth", &pd));
  VKR_TRY(make_tex(current_ao, 0, VKR_FMT_R16_SFLOAT, "gtao_reproject.current_ao", &cur));
  VKR_TRY(make_tex(prev_ao, 0, VKR_FMT_R16_SFLOAT, "gtao_reproject.prev_ao", &prev));
  VKR_TRY(make_tex(out_img, 0, VKR_FMT_R16_SFLOAT, "gtao_reproject.out", &out));
  Proj pr;
  fill_proj(pr, params->fovy, params->aspect, params->znear, params->zfar);
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_gtao_reproject, grid2d(out.w, out.h, block), block, 0, (hipStream_t)stream, d, pd, cur, prev, out, pr,
                     (out.fw / 8) * 8, (out.fh / 4) * 4);
  return launch_status("gtao_reproject");
}

extern "C" int vkr_deinterleave_depth(const vkr_img* depth, const vkr_img* layers, uint32_t layer_count,
                                      const vkr_deinterleave_push* push, void* stream) {
  if (!push) { set_error("deinterleave_depth: NULL push constants"); return VKR_ERR_NULL; }
  if (push->pattern_step < 0 || push->pattern_step > 3) { set_error("deinterleave_depth: pattern_step %d outside 0..3", push->pattern_step); return VKR_ERR_EXTENT; }
  Tex d, first;
  LayerSet set;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "deinterleave_depth.depth", &d));
  VKR_TRY(make_layers(layers, layer_count, "deinterleave_depth.out", &first, &set));
  const int tw = (first.fw / 8) * 8, th = (first.fh / 4) * 4;
  if (tw == 0 || th == 0) return VKR_OK;
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_deinterleave, grid2d(tw, th, block), block, 0, (hipStream_t)stream, d, set, push->pattern_step, tw, th);
  return launch_status("deinterleave_depth");
}

extern "C" int vkr_gtao_main_deinterleaved(const vkr_img* layers, uint32_t layer_count, const vkr_gtao_params* params,
                                           const vkr_img* normal, const vkr_img* out_raw,
                                           const vkr_gtao_deinterleaved_push* push, void* stream) {
  if (!params || !push) { set_error("main_deinterleaved: NULL params"); return VKR_ERR_NULL; }
  if (push->pattern_n < 0 || push->pattern_n > 3) { set_error("main_deinterleaved: pattern_n %d outside 0..3", push->pattern_n); return VKR_ERR_EXTENT; }
  Gtao2Args a;
  Tex first;
  LayerSet set;
  VKR_TRY(make_layers(layers, layer_count, "main_deinterleaved.depth_array", &first, &set));
  const uint32_t layer = push->layer < layer_count ? push->layer : layer_count - 1;  // array layer clamps
  VKR_TRY(make_tex(&layers[layer], 0, VKR_FMT_R32_SFLOAT, "main_deinterleaved.depth_array", &a.depth));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "main_deinterleaved.normal", &a.normal));
  VKR_TRY(make_tex(out_raw, 0, VKR_FMT_RGBA16_SFLOAT, "main_deinterleaved.out", &a.out));
  load_mat(a.normal_mat, params->normal_mat);
  fill_proj(a.pr, params->fovy, params->aspect, params->znear, params->zfar);
  fill_slice_table(a.slice_cs, push->angle_offset);
  const int scale = 1 << push->pattern_n;
  a.inv_w = (a.out.fw / 8) * 8; a.inv_h = (a.out.fh / 4) * 4;
  a.scale = scale;
  a.off_x = (int)(push->layer & (uint32_t)(scale - 1));
  a.off_y = (int)((push->layer >> push->pattern_n) & (uint32_t)(scale - 1));
  a.tex_w = scale * a.inv_w; a.tex_h = scale * a.inv_h;
  a.half_texel = 0;
  if (a.inv_w == 0 || a.inv_h == 0) return VKR_OK;
  // only invocations whose scaled position lands inside `out` do anything
  const int live_w = (a.out.fw - a.off_x + scale - 1) / scale, live_h = (a.out.fh - a.off_y + scale - 1) / scale;
  a.inv_w = a.inv_w < live_w ? a.inv_w : live_w;
  a.inv_h = a.inv_h < live_h ? a.inv_h : live_h;
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_gtao_v2<FmtR32F>, grid2d(a.inv_w, a.inv_h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("main_deinterleaved");
}

extern "C" int vkr_screen_trace_main(const vkr_img* depth, const vkr_img* normal, const vkr_img* color, const vkr_img* material,
                                     const vkr_img* out_raw, const vkr_screen_trace_params* params, void* stream) {
  if (!params) { set_error("screen_trace_main: NULL params"); return VKR_ERR_NULL; }
  ScreenTraceArgs a;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "screen_trace_main.depth", &a.depth));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "screen_trace_main.normal", &a.normal));
  VKR_TRY(make_tex(color, 0, VKR_FMT_RGBA8_SRGB, "screen_trace_main.color", &a.color));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "screen_trace_main.material", &a.material));
  VKR_TRY(make_tex(out_raw, 0, VKR_FMT_RGBA16_SFLOAT, "screen_trace_main.out", &a.out));
  if (a.out.ox != 0 || a.out.oy != 0 || a.out.w != a.out.fw || a.out.h != a.out.fh) {
    set_error("screen_trace_main: tiled windows are not supported (8x8 tile exchange)");
    return VKR_ERR_EXTENT;
  }
  load_mat(a.normal_mat, params->normal_mat);
  fill_proj(a.pr, params->fovy, params->aspect, params->znear, params->zfar);
  fill_slice_table(a.slice_cs, params->angle_offset);
  a.random_offset = params->random_offset;
  const int groups_x = a.out.fw / ST_TILE, groups_y = a.out.fh / ST_TILE;
  a.tex_w = groups_x * ST_TILE; a.tex_h = groups_y * ST_TILE;
  if (groups_x == 0 || groups_y == 0) return VKR_OK;
  dim3 block(ST_TILE * ST_TILE * ST_TILES_X), grid((groups_x + ST_TILES_X - 1) / ST_TILES_X, groups_y);
  hipLaunchKernelGGL(k_screen_trace, grid, block, 0, (hipStream_t)stream, a);
  return launch_status("screen_trace_main");
}

extern "C" int vkr_screen_trace_filter(const vkr_img* raw, const vkr_img* depth, const vkr_img* out_filtered,
                                       const vkr_screen_trace_filter_push* push, void* stream) {
  if (!push) { set_error("screen_trace_filter: NULL push constants"); return VKR_ERR_NULL; }
  Tex r, d, out;
  VKR_TRY(make_tex(raw, 0, VKR_FMT_RGBA16_SFLOAT, "screen_trace_filter.raw", &r));
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "screen_trace_filter.depth", &d));
  VKR_TRY(make_tex(out_filtered, 0, VKR_FMT_RGBA16_SFLOAT, "screen_trace_filter.out", &out));
  dim3 block(SF_BX, SF_BY);
  hipLaunchKernelGGL(k_screen_trace_filter, grid2d(out.w, out.h, block), block, 0, (hipStream_t)stream, r, d, out,
                     (out.fw / 8) * 8, (out.fh / 4) * 4, push->znear, push->zfar);
  return launch_status("screen_trace_filter");
}

extern "C" int vkr_screen_trace_accumulate(const vkr_img* depth, const vkr_img* prev_depth, const vkr_img* current,
                                           const vkr_img* accum_inout, const vkr_screen_trace_accum_push* push, void* stream) {
  if (!push) { set_error("screen_trace_accumulate: NULL push constants"); return VKR_ERR_NULL; }
  Tex d, pd, cur, acc;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "screen_trace_accumulate.depth", &d));
  VKR_TRY(make_tex(prev_depth, 0, VKR_FMT_D24_UNORM_S8, "screen_trace_accumulate.prev_depth", &pd));
  VKR_TRY(make_tex(current, 0, VKR_FMT_RGBA16_SFLOAT, "screen_trace_accumulate.current", &cur));
  VKR_TRY(make_tex(accum_inout, 0, VKR_FMT_RGBA16_SFLOAT, "screen_trace_accumulate.accum", &acc));
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_screen_trace_accumulate, grid2d(acc.w, acc.h, block), block, 0, (hipStream_t)stream, d, pd, cur, acc,
                     (acc.fw / 8) * 8, (acc.fh / 4) * 4, push->znear, push->zfar);
  return launch_status("screen_trace_accumulate");
}
