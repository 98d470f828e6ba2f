// gtao.hip — programs "gtao_compute_main", "gtao_filter", "gtao_accumulate".
//
// Reference: src/gtao.cpp:84-148,198-239,286-347 and shaders/gtao/{main,filter,accum}.comp.
// All three run at half resolution (GTAO(.., half_res = 1), main.cpp:265).
// Roofline: HBM for filter/accumulate (3.5 / 5.5 B per full-res pixel); the main pass is
// 16 dependent bilinear depth taps + ~8 transcendentals per pixel against 13 B (MIS) or
// 7 B (non-MIS) and is latency/ALU-bound.
#include <cstdlib>
#include "vkr_host.hpp"

namespace vkr {

struct GtaoArgs {
  Tex depth, normal, material, pdf, out;
  Mat4 normal_mat;
  Proj pr;
  int tex_w, tex_h;          // floor-dispatch extent (main.comp:54, gtao.cpp:145)
  float slice_cs[2][16][2];  // [dir_index][gtao_direction*16] -> (cos, sin) of the slice angle, host libm
  float weight_ratio;
  uint32_t use_mis, two_directions, reflections_only;
};

// Depth tile staged in LDS for the horizon search.  A block resolves GT_BX x GT_BY pixels; every
// sample of find_horizon lies within min(100/|P|, 16) pixels of its pixel (main.comp:229) and the
// bilinear footprint adds one more, so an apron of GT_R = 17 texels covers all 16 taps.  The tile
// holds decoded depth for frame coordinates *after* clamp-to-edge, so a tap is four LDS reads.
#define GT_BX 64
#define GT_BY 16
#define GT_R 17
#define GT_TW (GT_BX + 2 * GT_R)
#define GT_TH (GT_BY + 2 * GT_R)

struct DepthTile {
  const float* d;
  int x0, y0;  // frame coordinates of tile texel (0,0)
  float x0f, y0f;  // the same as floats (|.| < 2^16: exact)
  float fw, fh;
};
// texture(depth, uv): same arithmetic as sample<FmtD24>, texels served from the tile.  The texel index is taken relative to
// the tile and clamped while still a float (floor values and the tile origin are integers far below 2^24, so the
// difference is exact; a NaN clamps to the low bound), then converted: one v_med3_f32 per axis instead of two integer ones.
VKR_DEV float tile_sample(const DepthTile& t, f2 uv) {
  float x = cfma(uv.x, t.fw, -0.5f), y = cfma(uv.y, t.fh, -0.5f);
  float x0f = floorf(x), y0f = floorf(y);
  float fx = x - x0f, fy = y - y0f;
  const int tx = (int)__builtin_amdgcn_fmed3f(x0f - t.x0f, 0.0f, (float)(GT_TW - 2));
  const int ty = (int)__builtin_amdgcn_fmed3f(y0f - t.y0f, 0.0f, (float)(GT_TH - 2));
  const float* p = t.d + (__umul24((uint32_t)ty, (uint32_t)GT_TW) + (uint32_t)tx);
  return mixf(mixf(p[0], p[1], fx), mixf(p[GT_TW], p[GT_TW + 1], fx), fy);
}

// texture(depth, uv) of the main pass.  TILED: from the LDS tile, valid when screen_uv maps pixel (x, y) onto
// texel (x, y), i.e. when the floor-dispatch extent equals the image extent (every size divisible by 8 x 4: all
// benchmark configurations).  Ragged sizes stretch the uv -> texel map by fw / tex_w (main.comp:54 derives tex_size
// from gl_NumWorkGroups), the samples of a block then leave its tile, and the pass reads global memory instead.
template <bool TILED> VKR_DEV float depth_sample(const DepthTile& t, const Tex& depth, f2 uv) {
  if (TILED) return tile_sample(t, uv);
  return sample<FmtD24>(depth, uv);
}

// main.comp:84-108
template <bool TILED>
VKR_DEV float find_horizon(const DepthTile& depth, const Tex& depth_tex, const Proj& pr, f2 start, f3 camera_start, f2 dir, f3 v) {
  float h_cos = -1.0f;
  float previous_z = camera_start.z;
  float s = 0.0f;  // i / 16: multiples of 2^-4 add exactly
#pragma unroll 1
  for (int i = 1; i <= 16; i++) {
    s += 0.0625f;
    f2 tc = madd(start, s, dir);
    float sample_depth = depth_sample<TILED>(depth, depth_tex, tc);
    f3 sample_pos = reconstruct_view_vec(tc, sample_depth, pr);
    if (sample_pos.z > previous_z + 0.1f) break;  // MAX_THIKNESS, main.comp:82
    previous_z = sample_pos.z;
    f3 sample_offset = sample_pos - camera_start;
    // max()-reduced cosine: the hardware rsq is accurate enough (the break test above stays exact)
    float sample_cos = dot(v, sample_offset) * fast_rsq(dot(sample_offset, sample_offset));
    asm("v_max_f32 %0, %1, %2" : "=v"(h_cos) : "v"(h_cos), "v"(sample_cos));  // fmaxf without the canonicalising self-max (a NaN cosine is dropped either way)
  }
  return h_cos;
}

// One thread per half-res pixel.  The slice-direction pattern repeats every 4x4 pixels
// (main.comp:276-278), so its 16 (cos,sin) pairs are kernel arguments evaluated once on the host
// instead of per pixel.
// MIS: use_mis as a compile-time constant (one slice, no loop-carried sum: fewer live values across the horizon search).
template <bool TILED, bool MIS>
// (a block is 16 waves: with more than 64 VGPRs only one block fits a CU — 4 waves per SIMD — and the pass is 11 % slower)
__global__ __launch_bounds__(GT_BX * GT_BY, 8) void k_gtao_main(GtaoArgs a) {
  const i2 blk = xcd_block<2, 4>();  // chunks of 128 x 64 output pixels
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  __shared__ float s_depth[GT_TW * GT_TH];
  const int tid = threadIdx.y * GT_BX + threadIdx.x;
  DepthTile tile;
  tile.d = s_depth;
  tile.x0 = a.out.ox + blk.x * GT_BX - GT_R;
  tile.y0 = a.out.oy + blk.y * GT_BY - GT_R;
  tile.x0f = (float)tile.x0; tile.y0f = (float)tile.y0;
  tile.fw = (float)a.depth.fw; tile.fh = (float)a.depth.fh;
  // every load of the staging phase is issued before the first is waited for: the table entry and the <= 5 tile texels of a
  // thread (one wait instead of six in a row: a block is 16 of the CU's 32 waves, and they all stand at this barrier)
  constexpr int STAGE = (GT_TW * GT_TH + GT_BX * GT_BY - 1) / (GT_BX * GT_BY);
  uint32_t raw[STAGE];
  const uint32_t lut_bits = k_srgb_decode_bits[tid & (VKR_SRGB_LUT_SIZE - 1)];
  if (TILED) {
#pragma unroll
    for (int k = 0; k < STAGE; k++) {
      const int t = min(tid + k * GT_BX * GT_BY, GT_TW * GT_TH - 1);  // the last batch re-stages the last texel
      const int lx = iclamp(tile.x0 + t % GT_TW - a.depth.ox, 0, a.depth.w - 1), ly = iclamp(tile.y0 + t / GT_TW - a.depth.oy, 0, a.depth.h - 1);
      raw[k] = *(const uint32_t*)(a.depth.p + toff(a.depth, lx, ly, 4));
    }
  }
  if (tid < VKR_SRGB_LUT_SIZE) s_lut[tid] = __uint_as_float(lut_bits);
  if (TILED) {
#pragma unroll
    for (int k = 0; k < STAGE; k++) s_depth[min(tid + k * GT_BX * GT_BY, GT_TW * GT_TH - 1)] = FmtD24::decode(raw[k]);
  }
  __syncthreads();
  const int lx = blk.x * GT_BX + threadIdx.x;
  const int ly = blk.y * GT_BY + threadIdx.y;
  if (lx >= a.out.w || ly >= a.out.h) return;
  const int gx = a.out.ox + lx, gy = a.out.oy + ly;
  if (gx >= a.tex_w || gy >= a.tex_h) return;
  const f2 screen_uv = mk2(pixel_centre_uv(gx, (float)a.tex_w), pixel_centre_uv(gy, (float)a.tex_h));
  const float pdf_uniform = 1.0f / (2.0f * VKR_PI);
  float occ_x = 0.0f, occ_y = pdf_uniform;
  // the output texel as a 32-bit offset (a 64-bit pointer formed here would be live, two registers, across the horizon search)
  const uint32_t dst_off = toff(a.out, lx, ly, 8);

  const float frag_depth = depth_sample<TILED>(tile, a.depth, screen_uv);
  if (frag_depth >= 1.0f) {  // sky: mis -> (0,1), non-mis -> 0 (main.comp:187-189,221-223)
    occ_x = 0.0f;
    occ_y = MIS ? 1.0f : pdf_uniform;
  } else {
    // Exact: camera_pos, the sample radius / direction (they place the horizon samples and feed the
    // break test) and w0 / camera_normal / L (coordinates of the ill-conditioned PDF lookup).
    // Smooth (hardware rsq): the slice frame and its angles.
    const f3 camera_pos = reconstruct_view_vec(screen_uv, frag_depth, a.pr);
    const f3 w0 = -normalize(camera_pos);
    const f3 n_world = decode_normal(sample<FmtRG16U>(a.normal, screen_uv));
    const f3 camera_normal = normalize(xyz(mul(a.normal_mat, mk4(n_world.x, n_world.y, n_world.z, 0.0f))));
    const float rad = vmin(100.0f / length(camera_pos), 16.0f);
    const f2 dir_radius = mk2(rad / (float)a.depth.fw, rad / (float)a.depth.fh);
    const int dir_slot = (((gx + gy) & 3) << 2) + (gx & 3);  // 16 * gtao_direction(pos)
    const int dirs = MIS ? 1 : (a.two_directions ? 2 : 1);
    float sum = 0.0f, occlusion = 0.0f;
    f3 L = mk3(0, 0, 0);
    for (int di = 0; di < dirs; di++) {
      const f2 cs = mk2(a.slice_cs[di][dir_slot][0], a.slice_cs[di][dir_slot][1]);
      const f2 sample_direction = dir_radius * cs;
      const f3 sample_end_pos = reconstruct_view_vec(screen_uv + sample_direction, frag_depth, a.pr);
      f3 slice_normal = normalize_fast(cross(w0, -sample_end_pos));
      f3 normal_projected = madd(camera_normal, -dot(camera_normal, slice_normal), slice_normal);
      f3 X = -normalize_fast(cross(slice_normal, w0));
      const float np_len2 = dot(normal_projected, normal_projected);
      float np_len = fast_sqrt(np_len2);
      float n_cos = dot(normal_projected, X) * fast_rsq(np_len2);
      if (!(fabsf(n_cos) <= 0.9999f)) {
        // acos cliff (surface seen edge-on in this slice): whether the argument rounds past +-1 — NaN in the shader,
        // which zeroes the slice's arc — is decided by its last bit, so this rare case takes the exact sequence
        slice_normal = normalize(cross(w0, -sample_end_pos));
        normal_projected = madd(camera_normal, -dot(camera_normal, slice_normal), slice_normal);
        X = -normalize(cross(slice_normal, w0));
        n_cos = dot(normalize(normal_projected), X);
        np_len = length(normal_projected);
      }
      const float n = VKR_PI / 2.0f - acosf(n_cos);
      const float h_cos = find_horizon<TILED>(tile, a.depth, a.pr, screen_uv, camera_pos, sample_direction, w0);
      float h = acosf(h_cos);
      h = vmin(n + vmin(h - n, VKR_PI / 2.0f), h);
      const float arc = vmax((-cosf(2.0f * h - n) + cosf(n)) + (2.0f * h) * sinf(n), 0.0f);
      if (MIS) {
        occlusion = (((1.0f / VKR_PI) * np_len) * 0.25f) * arc;
        L = normalize(sample_end_pos - camera_pos);
      } else {
        sum += (np_len * 0.25f) * arc;
      }
    }
    if (!MIS) {
      occ_x = (2.0f * sum) / (float)dirs;  // main.comp:216
    } else {
      // main.comp:250-273
      const float roughness = sample_srgb_channel(a.material, screen_uv, 1, s_lut);
      const float pdf_ggx = sampleGGXdirPDF(a.pdf, w0, camera_normal, L, roughness * roughness);
      const uint2 prev = *(const uint2*)(a.out.p + dst_off);  // imageLoad(gtao_out): (occlusion, pdf) written by the SSR trace
      const float ao_x = half_bits_to_float(prev.x & 0xFFFFu), ao_y = half_bits_to_float(prev.x >> 16);
      if (a.reflections_only != 0) {
        float res = ao_x / ao_y;
        occ_x = is_nan(res) ? 1.0f : res;
        occ_y = 1.0f;
      } else {
        const float alpha = 1.0f / (a.weight_ratio + 1.0f);
        const float betta = 1.0f - alpha;
        const float mis_weight1 = alpha * fast_rcp(alpha * ao_y + betta * pdf_uniform);
        const float mis_weight2 = betta * fast_rcp(alpha * pdf_ggx + betta * pdf_uniform);
        const float mis_ao = ao_x * mis_weight1 + occlusion * mis_weight2;
        occ_x = is_nan(mis_ao) ? occlusion / pdf_uniform : mis_ao;
        occ_y = 1.0f;
      }
    }
  }
  uint2 o;
  o.x = float_to_half_bits(occ_x) | (float_to_half_bits(occ_y) << 16);
  o.y = 0u;
  *(uint2*)(const_cast<uint8_t*>(a.out.p) + dst_off) = o;
}

// filter.comp:17-51: 4x4 taps at offsets -2..+1, depth-weighted mean of raw.r.  The block stages
// {linearised depth, raw.r} of its pixels plus the (-2..+1) apron in LDS; out-of-frame taps read 0
// like the shader's texelFetch (linearize(0) = -znear, raw = 0).
#define GF_BX 64
#define GF_BY 4
#define GF_TW (GF_BX + 3)
#define GF_TH (GF_BY + 3)
__global__ __launch_bounds__(GF_BX * GF_BY) void k_gtao_filter(Tex depth, Tex raw, Tex out, int tex_w, int tex_h, float znear, float zfar) {
  const i2 blk = xcd_block<2, 16>();  // chunks of 128 x 64 output pixels
  __shared__ float2 s_t[GF_TW * GF_TH];
  const int tid = threadIdx.y * GF_BX + threadIdx.x;
  const int bx0 = out.ox + blk.x * GF_BX - 2, by0 = out.oy + blk.y * GF_BY - 2;
  for (int t = tid; t < GF_TW * GF_TH; t += GF_BX * GF_BY) {
    const int px = bx0 + t % GF_TW, py = by0 + t / GF_TW;
    s_t[t] = make_float2(linearize_depth2_unorm(fetch<FmtD24>(depth, px, py), znear, zfar), fetch<FmtRGBA16F>(raw, px, py).x);
  }
  __syncthreads();
  const int lx = blk.x * GF_BX + threadIdx.x;
  const int ly = blk.y * GF_BY + threadIdx.y;
  if (lx >= out.w || ly >= out.h) return;
  const int gx = out.ox + lx, gy = out.oy + ly;
  if (gx >= tex_w || gy >= tex_h) return;
  const int tc = (threadIdx.y + 2) * GF_TW + (threadIdx.x + 2);
  const float linear_depth = s_t[tc].x;
  // the tap weight max(0, 1 - 5 |dz| / |z|) is a continuous factor of a fp16 result: one hardware
  // reciprocal of |z| serves the 16 taps (the shader's 16 divisions are 30 % of this kernel otherwise)
  const float k = 5.0f * fast_rcp(fabsf(linear_depth));
  float weight_sum = 0.0f, ao = 0.0f;
#pragma unroll
  for (int x = 0; x < 4; x++) {
#pragma unroll
    for (int y = 0; y < 4; y++) {
      const float2 sm = s_t[tc + (x - 2) + (y - 2) * GF_TW];
      const float weight = vmax(0.0f, __builtin_fmaf(-fabsf(sm.x - linear_depth), k, 1.0f));
      weight_sum += weight;
      ao += weight * sm.y;
    }
  }
  ao /= weight_sum;
  *texel_ptr<uint16_t>(out, lx, ly) = (uint16_t)float_to_half_bits(ao);
}

struct AccumArgs {
  Tex depth, prev_depth, cur_ao, out, velocity, history;
  Mat4 prev_inverse_camera, mvp;
  Proj pr;
  uint32_t clear_history;
};

// accum.comp:29-88
__global__ __launch_bounds__(256) void k_gtao_accumulate(AccumArgs a) {
  const i2 blk = xcd_block<2, 16>();  // chunks of 128 x 64 output pixels
  const int lx = blk.x * blockDim.x + threadIdx.x;
  const int ly = blk.y * blockDim.y + threadIdx.y;
  if (lx >= a.out.w || ly >= a.out.h) return;
  const int gx = a.out.ox + lx, gy = a.out.oy + ly;
  const f2 tex_size = mk2((float)a.out.fw, (float)a.out.fh);
  const f2 screen_uv = mk2(pixel_centre_uv(gx, tex_size.x), pixel_centre_uv(gy, tex_size.y));
  const f2 velocity = sample<FmtRG16F>(a.velocity, screen_uv);
  const f2 prev_uv = screen_uv + velocity;
  bool reprojected = false;
  float valid_samples = 1.0f;
  if (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f) {
    const float dprev = sample<FmtD24>(a.prev_depth, prev_uv);
    const f3 vc = reconstruct_view_vec(prev_uv, dprev, a.pr);
    const f3 v_world_prev = xyz(mul(a.prev_inverse_camera, mk4(vc.x, vc.y, vc.z, 1.0f)));
    f4 prev_ndc = mul(a.mvp, mk4(v_world_prev.x, v_world_prev.y, v_world_prev.z, 1.0f));
    prev_ndc = prev_ndc / prev_ndc.w;
    const f2 prev_world_uv = mk2(0.5f * prev_ndc.x + 0.5f, 0.5f * prev_ndc.y + 0.5f);
    const f2 delta = mk2(fabsf(prev_world_uv.x - screen_uv.x) * tex_size.x, fabsf(prev_world_uv.y - screen_uv.y) * tex_size.y);
    const float current_z = linearize_depth2_unorm(sample<FmtD24>(a.depth, screen_uv), a.pr.znear, a.pr.zfar);
    const float prev_z = linearize_depth2(prev_ndc.z, a.pr.znear, a.pr.zfar);
    const float depth_err = fabsf(prev_z - current_z);
    const float vel_delta = vmax(fabsf(velocity.x) * tex_size.x, fabsf(velocity.y) * tex_size.y);
    const float error = 0.1f * vel_delta + depth_err;
    valid_samples = vclamp(1.0f - error, 0.8f, 1.0f);
    reprojected = (vmax(delta.x, delta.y) <= 2.0f) && (depth_err < 0.2f);
  }
  const float new_ao = fetch<FmtR16F>(a.cur_ao, gx, gy);
  float computed_ao = new_ao;
  float samples_count = 1.0f;
  if (a.clear_history != 0) reprojected = false;
  if (reprojected) {
    const f2 accumulated = sample<FmtRG16F>(a.history, prev_uv);
    samples_count = (255.0f * accumulated.y) * valid_samples;
    computed_ao = (accumulated.x * samples_count + new_ao) / (samples_count + 1.0f);
    samples_count += 1.0f;
    if (samples_count > 255.0f) samples_count = 100.0f;
  }
  *texel_ptr<uint32_t>(a.out, lx, ly) =
      float_to_half_bits(vclamp(computed_ao, 0.0f, 1.0f)) | (float_to_half_bits(samples_count / 255.0f) << 16);
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_gtao_main(const vkr_img* depth, const vkr_gtao_params* params, const vkr_img* normal,
                             const vkr_img* material, const vkr_img* pdf_tex, const vkr_img* gtao_inout,
                             const vkr_gtao_push* push, void* stream) {
  if (!params || !push) { set_error("gtao_main: NULL params"); return VKR_ERR_NULL; }
  GtaoArgs a;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "gtao_main.depth", &a.depth));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "gtao_main.normal", &a.normal));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "gtao_main.material", &a.material));
  VKR_TRY(make_tex(pdf_tex, 0, VKR_FMT_R32_SFLOAT, "gtao_main.pdf", &a.pdf));
  VKR_TRY(make_tex(gtao_inout, 0, VKR_FMT_RGBA16_SFLOAT, "gtao_main.gtao_out", &a.out));
  load_mat(a.normal_mat, params->normal_mat);
  a.pr.tg = tanf(params->fovy / 2.0f);
  a.pr.aspect = params->aspect; a.pr.znear = params->znear; a.pr.zfar = params->zfar;
  a.tex_w = (a.out.fw / 8) * 8;
  a.tex_h = (a.out.fh / 4) * 4;
  a.weight_ratio = push->weight_ratio;
  a.use_mis = push->use_mis > 0 ? 1u : 0u;
  a.two_directions = push->two_directions;
  a.reflections_only = push->reflections_only;
  const float PI = 3.1415926535897932384626433832795f;
  const uint32_t dirs = a.use_mis ? 1u : (push->two_directions != 0 ? 2u : 1u);
  for (uint32_t di = 0; di < 2; di++) {
    for (int k = 0; k < 16; k++) {
      // main.comp:198,233: angle = 2*PI*(gtao_direction + angle_offset [+ dir_index/dirs_count])
      float base_angle = (1.0f / 16.0f) * (float)k + push->angle_offset;
      float angle = a.use_mis ? (2.0f * PI) * base_angle : (2.0f * PI) * (base_angle + (float)di / (float)dirs);
      a.slice_cs[di][k][0] = cosf(angle);
      a.slice_cs[di][k][1] = sinf(angle);
    }
  }
  dim3 block(GT_BX, GT_BY);
  // (the LDS tile is staged from frame coordinates clamped to the depth window, as sample<>() clamps: the output window may be
  // any part of the frame — a strip's own rows of a window image, host/frame.cpp — as long as both describe the same frame)
  const bool tiled = a.tex_w == a.out.fw && a.tex_h == a.out.fh && a.depth.fw == a.out.fw && a.depth.fh == a.out.fh;
  void (*kernel)(GtaoArgs) = tiled ? (a.use_mis ? k_gtao_main<true, true> : k_gtao_main<true, false>)
                                   : (a.use_mis ? k_gtao_main<false, true> : k_gtao_main<false, false>);
  hipLaunchKernelGGL(kernel, grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("gtao_main");
}

extern "C" int vkr_gtao_filter(const vkr_img* depth, const vkr_img* raw_gtao, const vkr_img* out_filtered,
                               const vkr_gtao_filter_push* push, void* stream) {
  if (!push) { set_error("gtao_filter: NULL push constants"); return VKR_ERR_NULL; }
  Tex d, raw, out;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "gtao_filter.depth", &d));
  VKR_TRY(make_tex(raw_gtao, 0, VKR_FMT_RGBA16_SFLOAT, "gtao_filter.raw", &raw));
  VKR_TRY(make_tex(out_filtered, 0, VKR_FMT_R16_SFLOAT, "gtao_filter.out", &out));
  dim3 block(GF_BX, GF_BY);
  hipLaunchKernelGGL(k_gtao_filter, grid2d(out.w, out.h, block), block, 0, (hipStream_t)stream, d, raw, out,
                     (out.fw / 8) * 8, (out.fh / 4) * 4, push->znear, push->zfar);
  return launch_status("gtao_filter");
}

extern "C" int vkr_gtao_accumulate(const vkr_img* depth, const vkr_img* prev_depth, const vkr_img* current_ao,
                                   const vkr_img* out_accumulated, const vkr_img* velocity, const vkr_img* history,
                                   const vkr_gtao_accum_params* params, const vkr_gtao_accum_push* push, void* stream) {
  if (!params || !push) { set_error("gtao_accumulate: NULL params"); return VKR_ERR_NULL; }
  AccumArgs a;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "gtao_accumulate.depth", &a.depth));
  VKR_TRY(make_tex(prev_depth, 0, VKR_FMT_D24_UNORM_S8, "gtao_accumulate.prev_depth", &a.prev_depth));
  VKR_TRY(make_tex(current_ao, 0, VKR_FMT_R16_SFLOAT, "gtao_accumulate.current_ao", &a.cur_ao));
  VKR_TRY(make_tex(out_accumulated, 0, VKR_FMT_RG16_SFLOAT, "gtao_accumulate.out", &a.out));
  VKR_TRY(make_tex(velocity, 0, VKR_FMT_RG16_SFLOAT, "gtao_accumulate.velocity", &a.velocity));
  VKR_TRY(make_tex(history, 0, VKR_FMT_RG16_SFLOAT, "gtao_accumulate.history", &a.history));
  load_mat(a.prev_inverse_camera, params->prev_inverse_camera);
  load_mat(a.mvp, params->mvp);
  a.pr.tg = tanf(params->fovy_aspect_znear_zfar[0] / 2.0f);
  a.pr.aspect = params->fovy_aspect_znear_zfar[1];
  a.pr.znear = params->fovy_aspect_znear_zfar[2];
  a.pr.zfar = params->fovy_aspect_znear_zfar[3];
  a.clear_history = push->clear_history;
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_gtao_accumulate, grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("gtao_accumulate");
}
