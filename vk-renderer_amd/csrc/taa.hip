// taa.hip — program "taa_resolve".
//
// Reference: src/taa.cpp:19-63 + shaders/taa/resolve.comp:20-77.  Full resolution;
// the largest byte mover of the chain: 32 B per pixel (history 8 + prev_depth 4 + depth 4 +
// velocity 4 + colour 4 read, target 8 written).  Roofline: HBM.
#include "vkr_host.hpp"

namespace vkr {

struct TaaArgs {
  Tex history, hist_depth, cur_depth, velocity, color, out;
  Mat4 inverse_camera, prev_inverse_camera;
  Proj pr;
};

VKR_DEV f3 rgb(f4 v) { return mk3(v.x, v.y, v.z); }

__global__ __launch_bounds__(256) void k_taa_resolve(TaaArgs a) {
  const i2 blk = xcd_block<2, 16>();  // chunks of 128 x 64 output pixels
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  srgb_lut_stage(s_lut, threadIdx.y * blockDim.x + threadIdx.x, 256);
  __syncthreads();
  const int lx = blk.x * blockDim.x + threadIdx.x;
  const int ly = blk.y * blockDim.y + threadIdx.y;
  if (lx >= a.out.w || ly >= a.out.h) return;
  const int gx = a.out.ox + lx, gy = a.out.oy + ly;
  const f2 screen_uv = mk2(((float)gx + 0.5f) / (float)a.out.fw, ((float)gy + 0.5f) / (float)a.out.fh);
  const f3 current_color = sample_srgb_rgb(a.color, screen_uv, s_lut);
  const f2 velocity = sample<FmtRG16F>(a.velocity, screen_uv);
  const float delta_len = length(velocity);
  const f2 prev_uv = screen_uv + velocity;
  f3 out_color = current_color;
  if (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f) {
    // texture(prev_uv) and its four textureOffset neighbours (resolve.comp:41-45) share weights and
    // overlap in texels: the 12 distinct texels of the plus-shaped footprint are loaded and decoded
    // once; each of the five results is then the same lerp-of-lerps the sampler would compute.
    const float hx = cfma(prev_uv.x, (float)a.history.fw, -0.5f), hy = cfma(prev_uv.y, (float)a.history.fh, -0.5f);
    const float hx0f = floorf(hx), hy0f = floorf(hy);
    const float fx = hx - hx0f, fy = hy - hy0f;
    const int hx0 = f2i(hx0f), hy0 = f2i(hy0f);
    auto tex = [&](int dx, int dy) { return rgb(fetch_clamped<FmtRGBA16F>(a.history, hx0 + dx, hy0 + dy)); };
    const f3 t_m1_0 = tex(-1, 0), t_m1_1 = tex(-1, 1);
    const f3 t_0_m1 = tex(0, -1), t_0_0 = tex(0, 0), t_0_1 = tex(0, 1), t_0_2 = tex(0, 2);
    const f3 t_1_m1 = tex(1, -1), t_1_0 = tex(1, 0), t_1_1 = tex(1, 1), t_1_2 = tex(1, 2);
    const f3 t_2_0 = tex(2, 0), t_2_1 = tex(2, 1);
    auto bil = [&](f3 t00, f3 t10, f3 t01, f3 t11) { return mix3(mix3(t00, t10, fx), mix3(t01, t11, fx), fy); };
    f3 history = bil(t_0_0, t_1_0, t_0_1, t_1_1);
    const f3 color0 = bil(t_1_0, t_2_0, t_1_1, t_2_1);    // offset (+1, 0)
    const f3 color1 = bil(t_0_1, t_1_1, t_0_2, t_1_2);    // offset ( 0,+1)
    const f3 color2 = bil(t_m1_0, t_0_0, t_m1_1, t_0_1);  // offset (-1, 0)
    const f3 color3 = bil(t_0_m1, t_1_m1, t_0_0, t_1_0);  // offset ( 0,-1)
    const f3 color_min = min3(color0, min3(color1, min3(color2, color3)));
    const f3 color_max = max3(color0, max3(color1, max3(color2, color3)));
    history = min3(max3(history, color_min), color_max);
    const f3 blended = mix3(history, current_color, 0.1f);
    const f3 vc = reconstruct_view_vec(screen_uv, sample<FmtD24>(a.cur_depth, screen_uv), a.pr);
    const f3 v_world_cur = xyz(mul(a.inverse_camera, mk4(vc.x, vc.y, vc.z, 1.0f)));
    const f3 vp = reconstruct_view_vec(prev_uv, sample<FmtD24>(a.hist_depth, prev_uv), a.pr);
    const f3 v_world_prev = xyz(mul(a.prev_inverse_camera, mk4(vp.x, vp.y, vp.z, 1.0f)));
    const f3 v_camera = xyz(mul(a.inverse_camera, mk4(0, 0, 0, 1)));
    const float error = length(v_world_cur - v_world_prev);
    const float pixel_dist = length(v_world_cur - v_camera);
    const bool reprojected = (delta_len < 0.005f) || (error < vclamp((0.1f * pixel_dist) * delta_len, 0.01f, 0.2f));
    if (reprojected) out_color = blended;
  }
  uint2 o;
  o.x = float_to_half_bits(out_color.x) | (float_to_half_bits(out_color.y) << 16);
  o.y = float_to_half_bits(out_color.z);  // alpha 0 (resolve.comp:69)
  *texel_ptr<uint2>(a.out, lx, ly) = o;
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_taa_resolve(const vkr_img* history_color, const vkr_img* history_depth, const vkr_img* current_depth,
                               const vkr_img* velocity, const vkr_img* color, const vkr_img* out_color,
                               const vkr_reproject_params* params, void* stream) {
  if (!params) { set_error("taa_resolve: NULL params"); return VKR_ERR_NULL; }
  TaaArgs a;
  VKR_TRY(make_tex(history_color, 0, VKR_FMT_RGBA16_SFLOAT, "taa_resolve.history", &a.history));
  VKR_TRY(make_tex(history_depth, 0, VKR_FMT_D24_UNORM_S8, "taa_resolve.history_depth", &a.hist_depth));
  VKR_TRY(make_tex(current_depth, 0, VKR_FMT_D24_UNORM_S8, "taa_resolve.current_depth", &a.cur_depth));
  VKR_TRY(make_tex(velocity, 0, VKR_FMT_RG16_SFLOAT, "taa_resolve.velocity", &a.velocity));
  VKR_TRY(make_tex(color, 0, VKR_FMT_RGBA8_SRGB, "taa_resolve.color", &a.color));
  VKR_TRY(make_tex(out_color, 0, VKR_FMT_RGBA16_SFLOAT, "taa_resolve.out", &a.out));
  load_mat(a.inverse_camera, params->inverse_camera);
  load_mat(a.prev_inverse_camera, params->prev_inverse_camera);
  a.pr.tg = tanf(params->fovy_aspect_znear_zfar[0] / 2.0f);
  a.pr.aspect = params->fovy_aspect_znear_zfar[1];
  a.pr.znear = params->fovy_aspect_znear_zfar[2];
  a.pr.zfar = params->fovy_aspect_znear_zfar[3];
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_taa_resolve, grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("taa_resolve");
}
