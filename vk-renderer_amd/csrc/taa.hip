// taa.hip — program "taa_resolve".
//
// Reference: src/taa.cpp:19-63 + shaders/taa/resolve.comp:20-77.  Full resolution;
// the largest byte mover of the chain: 32 B per pixel (history 8 + prev_depth 4 + depth 4 +
// velocity 4 + colour 4 read, target 8 written).  Roofline: HBM in the compulsory model; in practice the pass sits
// between its VALU and texture-address limits (DESIGN.md section 3 "TAA"), hence the shared footprints and wide loads.
#include <cstdlib>
#include "vkr_host.hpp"

namespace vkr {

struct TaaArgs {
  Tex history, hist_depth, cur_depth, velocity, color, out;
  Mat4 inverse_camera, prev_inverse_camera;
  Proj pr;
  float still_d2;  // smallest d2 with sqrtf(d2) >= 0.005f: |velocity| < 0.005 <=> dot(velocity, velocity) < still_d2
};

// smallest float x with sqrtf(x) >= limit (correctly rounded sqrt is monotone)
inline float sqrt_threshold(float limit) {
  float x = limit * limit;
  while (sqrtf(x) >= limit) x = nextafterf(x, 0.0f);
  while (sqrtf(x) < limit) x = nextafterf(x, 1.0f);
  return x;
}

// Byte offsets (4-byte texels) and weights of the bilinear footprint of texture(t, uv): what sample<F>() computes
// before it loads.  Images that share one window geometry and pitch share the whole record.
struct Footprint { uint32_t o00, o10, o01, o11; float fx, fy; };
VKR_DEV Footprint footprint4(const Tex& t, f2 uv) {
  Footprint f;
  const float x = cfma(uv.x, (float)t.fw, -0.5f), y = cfma(uv.y, (float)t.fh, -0.5f);
  const float x0f = floorf(x), y0f = floorf(y);
  f.fx = x - x0f; f.fy = y - y0f;
  const int x0 = f2i(x0f) - t.ox, y0 = f2i(y0f) - t.oy;
  const uint32_t xa = (uint32_t)iclamp(x0, 0, t.w - 1) * 4u, xb = (uint32_t)iclamp(x0 + 1, 0, t.w - 1) * 4u;
  const uint32_t ra = __umul24((uint32_t)iclamp(y0, 0, t.h - 1), (uint32_t)t.pitch);
  const uint32_t rb = __umul24((uint32_t)iclamp(y0 + 1, 0, t.h - 1), (uint32_t)t.pitch);
  f.o00 = ra + xa; f.o10 = ra + xb; f.o01 = rb + xa; f.o11 = rb + xb;
  return f;
}
VKR_DEV BilinearTaps taps_at(const Tex& t, const Footprint& f) {
  BilinearTaps b;
  b.t00 = *(const uint32_t*)(t.p + f.o00); b.t10 = *(const uint32_t*)(t.p + f.o10);
  b.t01 = *(const uint32_t*)(t.p + f.o01); b.t11 = *(const uint32_t*)(t.p + f.o11);
  b.fx = f.fx; b.fy = f.fy;
  return b;
}

// SHARED: colour, velocity and current depth have one window geometry and pitch (the launcher checks), so the three
// samples at screen_uv share one footprint.
// TILED (implies SHARED): the three images also have the output's full extent, so texture(., screen_uv) of pixel g lands on
// the texels {g - 1, g} x {g - 1, g} or {g, g + 1} x {g, g + 1} (pixel-centre uv: uv * size - 0.5 is g up to 3e-4).  The
// 66 x 6 texels a 64 x 4 block can touch are then staged once per block in LDS — each loaded by one thread instead of by
// up to four, 1.5 instead of 4 loads per pixel and image — with the clamp-to-edge of the sampler applied while staging,
// and the taps read them there.  The pass is bound by the number of loads it issues (DESIGN_EXPERIMENTS.md A.8).
#ifndef TAA_BX
#define TAA_BX 64
#endif
#define TAA_BY (256 / TAA_BX)
#define TAA_TW (TAA_BX + 2)
#define TAA_TH (TAA_BY + 2)
#define TAA_STAGE ((TAA_TW * TAA_TH + 255) / 256)
template <bool SHARED, bool TILED>
__global__ __launch_bounds__(256) void k_taa_resolve(TaaArgs a) {
  const i2 blk = xcd_block<128 / TAA_BX, 64 / TAA_BY>();  // chunks of 128 x 64 output pixels
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  __shared__ uint32_t s_col[TILED ? TAA_TW * TAA_TH : 1], s_vel[TILED ? TAA_TW * TAA_TH : 1], s_dep[TILED ? TAA_TW * TAA_TH : 1];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  const int tile_x0 = a.out.ox + blk.x * TAA_BX - 1, tile_y0 = a.out.oy + blk.y * TAA_BY - 1;  // frame coordinates of tile texel (0, 0)
  if (TILED) {
    // both staging passes' loads (and the table's) are in flight together; texels past the tile's end re-stage its last one
    uint32_t off[TAA_STAGE], c[TAA_STAGE], v[TAA_STAGE], d[TAA_STAGE];
#pragma unroll
    for (int k = 0; k < TAA_STAGE; k++) {
      const int t = min(tid + k * 256, TAA_TW * TAA_TH - 1);
      const int cx = iclamp(tile_x0 + t % TAA_TW - a.color.ox, 0, a.color.w - 1), cy = iclamp(tile_y0 + t / TAA_TW - a.color.oy, 0, a.color.h - 1);
      off[k] = __umul24((uint32_t)cy, (uint32_t)a.color.pitch) + (uint32_t)cx * 4u;
    }
#pragma unroll
    for (int k = 0; k < TAA_STAGE; k++) {
      c[k] = *(const uint32_t*)(a.color.p + off[k]); v[k] = *(const uint32_t*)(a.velocity.p + off[k]); d[k] = *(const uint32_t*)(a.cur_depth.p + off[k]);
    }
    srgb_lut_stage(s_lut, tid, 256);
#pragma unroll
    for (int k = 0; k < TAA_STAGE; k++) {
      const int t = min(tid + k * 256, TAA_TW * TAA_TH - 1);
      s_col[t] = c[k]; s_vel[t] = v[k]; s_dep[t] = d[k];
    }
  } else {
    srgb_lut_stage(s_lut, tid, 256);
  }
  __syncthreads();
  const int lx = blk.x * blockDim.x + threadIdx.x;
  const int ly = blk.y * blockDim.y + threadIdx.y;
  if (lx >= a.out.w || ly >= a.out.h) return;
  const int gx = a.out.ox + lx, gy = a.out.oy + ly;
  // (g + 0.5) / size: both operands and the quotient are far inside the normal range
  const f2 screen_uv = mk2(pixel_centre_uv(gx, (float)a.out.fw), pixel_centre_uv(gy, (float)a.out.fh));
  Footprint fc, fv;
  BilinearTaps tc, tv, td_tile;
  if (TILED) {
    // the sampler's arithmetic (footprint4) up to the texel indices, which are taken relative to the tile; they cannot leave
    // it (see above) and are clamped into it all the same
    const float x = cfma(screen_uv.x, (float)a.color.fw, -0.5f), y = cfma(screen_uv.y, (float)a.color.fh, -0.5f);
    const float x0f = floorf(x), y0f = floorf(y);
    const int ix = iclamp(f2i(x0f) - tile_x0, 0, TAA_TW - 2), iy = iclamp(f2i(y0f) - tile_y0, 0, TAA_TH - 2);
    const int i00 = iy * TAA_TW + ix;
    tc.fx = tv.fx = td_tile.fx = x - x0f; tc.fy = tv.fy = td_tile.fy = y - y0f;
    tc.t00 = s_col[i00]; tc.t10 = s_col[i00 + 1]; tc.t01 = s_col[i00 + TAA_TW]; tc.t11 = s_col[i00 + TAA_TW + 1];
    tv.t00 = s_vel[i00]; tv.t10 = s_vel[i00 + 1]; tv.t01 = s_vel[i00 + TAA_TW]; tv.t11 = s_vel[i00 + TAA_TW + 1];
    td_tile.t00 = s_dep[i00]; td_tile.t10 = s_dep[i00 + 1]; td_tile.t01 = s_dep[i00 + TAA_TW]; td_tile.t11 = s_dep[i00 + TAA_TW + 1];
    fc = Footprint {}; fv = fc;
  } else {
    fc = footprint4(a.color, screen_uv);
    fv = SHARED ? fc : footprint4(a.velocity, screen_uv);
    tc = taps_at(a.color, fc); tv = taps_at(a.velocity, fv);
    td_tile = tc;  // (unused)
  }
  const f2 velocity = taps_resolve<FmtRG16F>(tv);
  const f2 prev_uv = screen_uv + velocity;
  const bool inside = prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f;
  // texture(prev_uv) and its four textureOffset neighbours (resolve.comp:41-45) share weights and
  // overlap in texels: the 12 distinct texels of the plus-shaped footprint are loaded and decoded
  // once; each of the five results is then the same lerp-of-lerps the sampler would compute.
  float fx = 0.0f, fy = 0.0f;
  uint2 h[12];
  if (inside) {
    const float hx = cfma(prev_uv.x, (float)a.history.fw, -0.5f), hy = cfma(prev_uv.y, (float)a.history.fh, -0.5f);
    const float hx0f = floorf(hx), hy0f = floorf(hy);
    fx = hx - hx0f; fy = hy - hy0f;
    const int hx0 = f2i(hx0f) - a.history.ox, hy0 = f2i(hy0f) - a.history.oy;
    if (hx0 >= 1 && hy0 >= 1 && hx0 + 2 < a.history.w && hy0 + 2 < a.history.h) {
      // interior footprint: pairs of horizontally adjacent texels as one 16-byte load each (6 loads instead of 12: measured
      // 0.094 -> 0.085 ms; pairing the 4-byte taps of the other footprints the same way gains nothing more)
      const uint8_t* p0 = a.history.p + (__umul24((uint32_t)(hy0 - 1), (uint32_t)a.history.pitch) + (uint32_t)(hx0 - 1) * 8u);
      const uint32_t pitch = (uint32_t)a.history.pitch;
      auto ld2 = [&](int ix, int iy) { return load_u32x4(p0 + ((uint32_t)(iy + 1) * pitch + (uint32_t)(ix + 1) * 8u)); };
      const U32x4 a_m1 = ld2(0, -1), a_2 = ld2(0, 2);                 // rows -1 and 2: columns 0, 1
      const U32x4 l_0 = ld2(-1, 0), r_0 = ld2(1, 0), l_1 = ld2(-1, 1), r_1 = ld2(1, 1);  // rows 0 and 1: columns -1, 0 and 1, 2
      h[0] = make_uint2(l_0.x, l_0.y); h[1] = make_uint2(l_1.x, l_1.y);
      h[2] = make_uint2(a_m1.x, a_m1.y); h[3] = make_uint2(l_0.z, l_0.w); h[4] = make_uint2(l_1.z, l_1.w); h[5] = make_uint2(a_2.x, a_2.y);
      h[6] = make_uint2(a_m1.z, a_m1.w); h[7] = make_uint2(r_0.x, r_0.y); h[8] = make_uint2(r_1.x, r_1.y); h[9] = make_uint2(a_2.z, a_2.w);
      h[10] = make_uint2(r_0.z, r_0.w); h[11] = make_uint2(r_1.z, r_1.w);
    } else {
      uint32_t xo[4], ro[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        xo[i] = (uint32_t)iclamp(hx0 + i - 1, 0, a.history.w - 1) * 8u;
        ro[i] = __umul24((uint32_t)iclamp(hy0 + i - 1, 0, a.history.h - 1), (uint32_t)a.history.pitch);
      }
      auto ld = [&](int ix, int iy) { return *(const uint2*)(a.history.p + (ro[iy + 1] + xo[ix + 1])); };
      h[0] = ld(-1, 0); h[1] = ld(-1, 1);
      h[2] = ld(0, -1); h[3] = ld(0, 0); h[4] = ld(0, 1); h[5] = ld(0, 2);
      h[6] = ld(1, -1); h[7] = ld(1, 0); h[8] = ld(1, 1); h[9] = ld(1, 2);
      h[10] = ld(2, 0); h[11] = ld(2, 1);
    }
  }
  const f3 current_color = mix3(mix3(srgb_rgb(tc.t00, s_lut), srgb_rgb(tc.t10, s_lut), tc.fx),
                                mix3(srgb_rgb(tc.t01, s_lut), srgb_rgb(tc.t11, s_lut), tc.fx), tc.fy);
  f3 out_color = current_color;
  if (inside) {
    // delta_len < 0.005 decided on the squared length (correctly rounded sqrt is monotone); the length itself is only
    // needed on the other side of the test
    const float d2 = dot(velocity, velocity);
    bool reprojected = d2 < a.still_d2;
    if (!reprojected) {
      const float delta_len = sqrtf(d2);
      float cur_depth;
      if (TILED) cur_depth = taps_resolve<FmtD24>(td_tile);
      else {
        const Footprint fd = SHARED ? fc : footprint4(a.cur_depth, screen_uv);
        cur_depth = taps_resolve<FmtD24>(taps_at(a.cur_depth, fd));
      }
      const f3 vc = reconstruct_view_vec(screen_uv, cur_depth, a.pr);
      const f3 v_world_cur = xyz(mul(a.inverse_camera, mk4(vc.x, vc.y, vc.z, 1.0f)));
      const f3 vp = reconstruct_view_vec(prev_uv, sample<FmtD24>(a.hist_depth, prev_uv), a.pr);
      const f3 v_world_prev = xyz(mul(a.prev_inverse_camera, mk4(vp.x, vp.y, vp.z, 1.0f)));
      const f3 v_camera = xyz(mul(a.inverse_camera, mk4(0, 0, 0, 1)));
      const float error = length(v_world_cur - v_world_prev);
      const float pixel_dist = length(v_world_cur - v_camera);
      reprojected = error < vclamp((0.1f * pixel_dist) * delta_len, 0.01f, 0.2f);
    }
    if (reprojected) {
      auto tex = [&](int i) { return mk3(half_bits_to_float(h[i].x & 0xFFFFu), half_bits_to_float(h[i].x >> 16), half_bits_to_float(h[i].y & 0xFFFFu)); };
      const f3 t_m1_0 = tex(0), t_m1_1 = tex(1);
      const f3 t_0_m1 = tex(2), t_0_0 = tex(3), t_0_1 = tex(4), t_0_2 = tex(5);
      const f3 t_1_m1 = tex(6), t_1_0 = tex(7), t_1_1 = tex(8), t_1_2 = tex(9);
      const f3 t_2_0 = tex(10), t_2_1 = tex(11);
      // the horizontal lerps of rows 0 and 1 between columns 0 and 1 serve three of the five samples
      const f3 r_m1 = mix3(t_0_m1, t_1_m1, fx), r_0 = mix3(t_0_0, t_1_0, fx), r_1 = mix3(t_0_1, t_1_1, fx), r_2 = mix3(t_0_2, t_1_2, fx);
      f3 history = mix3(r_0, r_1, fy);
      const f3 color0 = mix3(mix3(t_1_0, t_2_0, fx), mix3(t_1_1, t_2_1, fx), fy);    // offset (+1, 0)
      const f3 color1 = mix3(r_1, r_2, fy);                                           // offset ( 0,+1)
      const f3 color2 = mix3(mix3(t_m1_0, t_0_0, fx), mix3(t_m1_1, t_0_1, fx), fy);  // offset (-1, 0)
      const f3 color3 = mix3(r_m1, r_0, fy);                                          // offset ( 0,-1)
      const f3 color_min = min3(color0, min3(color1, min3(color2, color3)));
      const f3 color_max = max3(color0, max3(color1, max3(color2, color3)));
      history = min3(max3(history, color_min), color_max);
      out_color = mix3(history, current_color, 0.1f);
    }
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
  a.still_d2 = sqrt_threshold(0.005f);
  const bool shared = same_layout(a.color, a.velocity) && same_layout(a.color, a.cur_depth) && !(switches() & VKR_SWITCH_TAA_GENERIC);
  // the block's footprint tile needs texture(., screen_uv) to land next to the pixel: the images' full extent is the output's
  const bool tiled = shared && a.color.fw == a.out.fw && a.color.fh == a.out.fh && a.color.w >= 2 && a.color.h >= 2;
  dim3 block(TAA_BX, TAA_BY);
  if (tiled) hipLaunchKernelGGL((k_taa_resolve<true, true>), grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  else if (shared) hipLaunchKernelGGL((k_taa_resolve<true, false>), grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((k_taa_resolve<false, false>), grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("taa_resolve");
}
