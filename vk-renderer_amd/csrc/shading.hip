// shading.hip — programs "brdf_preintegrate" and "defered_shading" (SURVEY.md 8(f) #1): the step
// between GTAO/SSR and TAA that produces TAA's colour input in the reference frame loop
// (main.cpp:390-391).  Reference: src/defered_shading.cpp:47-118, shaders/defered_shading/shader.frag,
// src/advanced_ssr.cpp:116-136, shaders/advanced_ssr/preintegrate_ssr.comp.
// Full resolution, HBM-bound in the compulsory model: 23 B per pixel (albedo 4 + normal 4 + material 4
// + depth0 4 + depth1 1 + ao 1 + reflections 1 read, colour 4 written).
#include <cstdlib>
#include "vkr_host.hpp"

namespace vkr {

// brdf.glsl:135-155 with cos/sin(2*PI*U2) supplied (vkr_halton23_fill)
VKR_DEV f3 sampleGGXVNDF_cs(f3 Ve, float alpha_x, float alpha_y, float U1, float cos_phi, float sin_phi) {
  f3 Vh = normalize(mk3(alpha_x * Ve.x, alpha_y * Ve.y, Ve.z));
  float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
  f3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : mk3(1, 0, 0);
  f3 T2 = cross(Vh, T1);
  float r = sqrtf(U1);
  float t1 = r * cos_phi;
  float t2 = r * sin_phi;
  float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
  f3 Nh = (t1 * T1 + t2 * T2) + sqrtf(vmax(0.0f, (1.0f - t1 * t1) - t2 * t2)) * Vh;
  return normalize(mk3(alpha_x * Nh.x, alpha_y * Nh.y, vmax(0.0f, Nh.z)));
}

// preintegrate_ssr.comp:12-44 (one-time LUT; literal arithmetic)
__global__ __launch_bounds__(256) void k_brdf_preintegrate(const float4* __restrict__ halton, Tex out) {
  __shared__ float4 s_h[VKR_HALTON_SEQ_SIZE];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if (tid < VKR_HALTON_SEQ_SIZE) s_h[tid] = halton[tid];
  __syncthreads();
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= out.w || y >= out.h) return;
  const float roughness = ((float)x + 0.5f) / (float)out.fw;
  const float NdotV = ((float)y + 0.5f) / (float)out.fh;
  const float roughness2 = roughness * roughness;
  const f3 V = mk3(sqrtf(1.0f - NdotV * NdotV), 0.0f, NdotV);
  float A = 0.0f, B = 0.0f;
  for (int i = 0; i < VKR_HALTON_SEQ_SIZE; i++) {
    const float4 hv = s_h[i];
    const f3 H = sampleGGXVNDF_cs(V, roughness2, roughness2, hv.x, hv.z, hv.w);
    const f3 L = normalize(reflect(-V, H));
    const float NdotL = L.z;
    const float alpha = powf(1.0f - dot(V, H), 5.0f);
    const float G1 = brdfG1(roughness2, NdotV);
    const float G2 = brdfG2(NdotV, NdotL, roughness2);
    A += (G2 / G1) * (1.0f - alpha);
    B += (G2 / G1) * alpha;
  }
  A *= 1.0f / (float)VKR_HALTON_SEQ_SIZE;
  B *= 1.0f / (float)VKR_HALTON_SEQ_SIZE;
  *texel_ptr<uint32_t>(out, x, y) = float_to_half_bits(A) | (float_to_half_bits(B) << 16);
}

struct ShadingArgs {
  Tex albedo, normal, material, depth0, depth1, occlusion, brdf, reflections, out;
  Mat4 inverse_camera;
  Proj pr;
  float min_roughness, max_roughness;
  uint32_t show_ao;
};

// brdf.glsl:31-38
VKR_DEV float distribution_ggx(f3 N, f3 H, float alpha) {
  const float NoH = dot(N, H);
  const float alpha2 = alpha * alpha;
  const float NoH2 = NoH * NoH;
  const float den = NoH2 * alpha2 + (1.0f - NoH2);
  return (((NoH2 > 0.0f) ? 1.0f : 0.0f) * alpha2) / ((VKR_PI * den) * den);
}

// shader.frag:41-130.  The 2x2 nearest-depth pick (sample_ocllusion_ssr) compares exactly computed
// bilinear depths; the shading terms after it are smooth and written to an 8-bit sRGB target.
// FAST (the launcher checks): normal, albedo, material and depth share one window geometry and pitch, and every sampled
// image is at least two texels wide — the four full-resolution samples then share one footprint record, the material
// texels are fetched once for both channels, and every bilinear row is one 8-byte load: 19 loads per pixel instead of
// 41 (the pass is bound by the texture-address unit, which spends the same 16 cycles on a wave's load whatever its width).
template <bool FAST>
__global__ __launch_bounds__(256) void k_defered_shading(ShadingArgs a) {
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  __shared__ float s_thresh[VKR_SRGB_LUT_SIZE];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  srgb_lut_stage(s_lut, tid, 256);
  srgb_thresh_stage(s_thresh, tid, 256);
  __syncthreads();
  const int lx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ly = blockIdx.y * blockDim.y + threadIdx.y;
  if (lx >= a.out.w || ly >= a.out.h) return;
  const int gx = a.out.ox + lx, gy = a.out.oy + ly;
  const f2 screen_uv = mk2(pixel_centre_uv(gx, (float)a.out.fw), pixel_centre_uv(gy, (float)a.out.fh));
  // Exact: the depths that decide the 2x2 pick.  Everything after the pick is shading arithmetic
  // written to an 8-bit sRGB target: hardware rsq / rcp.
  f3 N, albedo;
  float roughness, metallic, depth;
  if (FAST) {
    const PairFootprint fp = pair_footprint(a.normal, screen_uv);
    const BilinearTaps tn = pair_taps(a.normal, fp), ta = pair_taps(a.albedo, fp), tm = pair_taps(a.material, fp), td = pair_taps(a.depth0, fp);
    N = decode_normal_fast(taps_resolve<FmtRG16U>(tn));
    albedo = mix3(mix3(srgb_rgb(ta.t00, s_lut), srgb_rgb(ta.t10, s_lut), ta.fx), mix3(srgb_rgb(ta.t01, s_lut), srgb_rgb(ta.t11, s_lut), ta.fx), ta.fy);
    roughness = taps_srgb_channel(tm, 1, s_lut);
    metallic = mixf(0.1f, 1.0f, taps_srgb_channel(tm, 2, s_lut));
    depth = taps_resolve<FmtD24>(td);
  } else {
    N = decode_normal_fast(sample<FmtRG16U>(a.normal, screen_uv));
    albedo = sample_srgb_rgb(a.albedo, screen_uv, s_lut);
    roughness = sample_srgb_channel(a.material, screen_uv, 1, s_lut);
    metallic = mixf(0.1f, 1.0f, sample_srgb_channel(a.material, screen_uv, 2, s_lut));
    depth = sample<FmtD24>(a.depth0, screen_uv);
  }
  // sample_ocllusion_ssr (:103-130): the four textureLodOffset taps share one 3x3 texel footprint
  float occlusion;
  f3 reflection;
  {
    const float hx = cfma(screen_uv.x, (float)a.depth1.fw, -0.5f), hy = cfma(screen_uv.y, (float)a.depth1.fh, -0.5f);
    const float hx0f = floorf(hx), hy0f = floorf(hy);
    const float fx = hx - hx0f, fy = hy - hy0f;
    const int x0 = f2i(hx0f), y0 = f2i(hy0f);
    float t[3][3];
    const int wx0 = x0 - a.depth1.ox, wy0 = y0 - a.depth1.oy;
    if (FAST && wx0 >= 0 && wy0 >= 0 && wx0 + 2 < a.depth1.w && wy0 + 2 < a.depth1.h) {
      // interior: a row of the footprint is the pair (x0, x0 + 1) as one 8-byte load plus texel x0 + 2
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const uint8_t* row = a.depth1.p + toff(a.depth1, wx0, wy0 + j, 4);
        const U32x2 pr = load_u32x2(row);
        t[j][0] = FmtD24::decode(pr.x); t[j][1] = FmtD24::decode(pr.y); t[j][2] = FmtD24::decode(*(const uint32_t*)(row + 8));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) t[j][i] = fetch_clamped<FmtD24>(a.depth1, x0 + i, y0 + j);
    }
    auto bil = [&](int ox, int oy) { return mixf(mixf(t[oy][ox], t[oy][ox + 1], fx), mixf(t[oy + 1][ox], t[oy + 1][ox + 1], fx), fy); };
    const float d0 = fabsf(bil(0, 0) - depth), d1 = fabsf(bil(1, 0) - depth), d2 = fabsf(bil(0, 1) - depth), d3 = fabsf(bil(1, 1) - depth);
    const float min_delta = vmin(vmin(d0, d1), vmin(d2, d3));
    int ox = 1, oy = 1;
    if (min_delta == d0) { ox = 0; oy = 0; }
    else if (min_delta == d1) { ox = 1; oy = 0; }
    else if (min_delta == d2) { ox = 0; oy = 1; }
    if (FAST) {
      occlusion = taps_resolve<FmtRG16F>(pair_taps(a.occlusion, pair_footprint(a.occlusion, screen_uv, ox, oy))).x;
      reflection = taps_resolve<FmtRGBA8>(pair_taps(a.reflections, pair_footprint(a.reflections, screen_uv, ox, oy)));
    } else {
      occlusion = sample<FmtRG16F>(a.occlusion, screen_uv, ox, oy).x;
      reflection = sample<FmtRGBA8>(a.reflections, screen_uv, ox, oy);
    }
  }
  const f3 vv = reconstruct_view_vec(screen_uv, depth, a.pr);
  const f3 world_pos = xyz(mul(a.inverse_camera, mk4(vv.x, vv.y, vv.z, 1.0f)));
  const f3 camera_pos = xyz(mul(a.inverse_camera, mk4(0, 0, 0, 1)));
  const f3 LIGHT_POS = mk3(-1.85867f, 5.81832f, -0.247114f);
  const f3 V = normalize_fast(camera_pos - world_pos);
  const f3 F0 = F0_approximation(albedo, metallic);
  const f3 Lv = LIGHT_POS - world_pos;
  const float ld2 = dot(Lv, Lv);
  const f3 L = Lv * fast_rsq(ld2);
  const f3 H = normalize_fast(V + L);
  const float rad = 0.1f * vmin(100.0f * fast_rcp(ld2), 100.0f);
  const float NdotL = vmax(dot(N, L), 0.0f), NdotV = vmax(dot(N, V), 0.0f);
  // DistributionGGX (brdf.glsl:31-38) and brdfG2 (:49-56) with hardware rcp / sqrt
  const float NoH = dot(N, H), NoH2 = NoH * NoH, alpha2 = roughness * roughness;
  const float den = NoH2 * alpha2 + (1.0f - NoH2);
  const float NDF = (NoH2 > 0.0f ? alpha2 : 0.0f) * fast_rcp((VKR_PI * den) * den);
  // G = brdfG2(NdotV, NdotL, roughness * roughness)
  const float NdotV2 = NdotV * NdotV, NdotL2 = NdotL * NdotL;
  const float L1 = fast_sqrt(1.0f + (alpha2 * (1.0f - NdotV2)) * fast_rcp(NdotV2));
  const float L2 = fast_sqrt(1.0f + (alpha2 * (1.0f - NdotL2)) * fast_rcp(NdotL2));
  const float G = 2.0f * fast_rcp(L1 + L2);
  const float om = vclamp(1.0f - vmax(dot(H, V), 0.0f), 0.0f, 1.0f), om2 = om * om;
  const f3 F = F0 + (mk3(1.0f, 1.0f, 1.0f) - F0) * ((om2 * om2) * om);
  const f3 kD = (mk3(1.0f, 1.0f, 1.0f) - F) * (1.0f - metallic);
  const f3 specular = ((NDF * G) * F) * fast_rcp((4.0f * NdotV) * NdotL + 0.0001f);
  const float biased_rougness = mixf(a.min_roughness, a.max_roughness, roughness);
  const f2 ssr_brdf = FAST ? taps_resolve<FmtRG16F>(pair_taps(a.brdf, pair_footprint(a.brdf, mk2(biased_rougness, NdotV))))
                           : sample<FmtRG16F>(a.brdf, mk2(biased_rougness, NdotV));
  f3 Lo = (((kD * albedo) * (1.0f / VKR_PI) + specular) * rad) * NdotL;
  Lo = Lo + reflection * (F0 * ssr_brdf.x + mk3(ssr_brdf.y, ssr_brdf.y, ssr_brdf.y));
  const f3 color = occlusion * (mk3(0.6f, 0.6f, 0.6f) * albedo + Lo);
  const f3 o = a.show_ao ? mk3(occlusion, occlusion, occlusion) : color;
  *texel_ptr<uint32_t>(a.out, lx, ly) =
      float_to_srgb8_lds(o.x, s_thresh) | (float_to_srgb8_lds(o.y, s_thresh) << 8) | (float_to_srgb8_lds(o.z, s_thresh) << 16);  // alpha 0
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_brdf_preintegrate(const float* halton_vec4, const vkr_img* out_brdf, void* stream) {
  if (!halton_vec4 || ((uintptr_t)halton_vec4 % 16) != 0) { set_error("brdf_preintegrate: halton buffer must be a 16-byte aligned device pointer"); return VKR_ERR_NULL; }
  Tex out;
  VKR_TRY(make_tex(out_brdf, 0, VKR_FMT_RG16_SFLOAT, "brdf_preintegrate.out", &out));
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_brdf_preintegrate, grid2d(out.w, out.h, block), block, 0, (hipStream_t)stream, (const float4*)halton_vec4, out);
  return launch_status("brdf_preintegrate");
}

extern "C" int vkr_defered_shading(const vkr_img* albedo, const vkr_img* normal, const vkr_img* material, const vkr_img* depth,
                                   const vkr_shading_params* consts, const vkr_img* occlusion, const vkr_img* brdf,
                                   const vkr_img* reflections, const vkr_img* out, const vkr_shading_push* push, void* stream) {
  if (!consts || !push) { set_error("defered_shading: NULL params"); return VKR_ERR_NULL; }
  ShadingArgs a;
  VKR_TRY(make_tex(albedo, 0, VKR_FMT_RGBA8_SRGB, "defered_shading.albedo", &a.albedo));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "defered_shading.normal", &a.normal));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "defered_shading.material", &a.material));
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "defered_shading.depth", &a.depth0));
  VKR_TRY(make_tex(depth, 1, VKR_FMT_D24_UNORM_S8, "defered_shading.depth (lod 1)", &a.depth1));
  VKR_TRY(make_tex(occlusion, 0, VKR_FMT_RG16_SFLOAT, "defered_shading.occlusion", &a.occlusion));
  VKR_TRY(make_tex(brdf, 0, VKR_FMT_RG16_SFLOAT, "defered_shading.brdf", &a.brdf));
  VKR_TRY(make_tex(reflections, 0, VKR_FMT_RGBA8_UNORM, "defered_shading.reflections", &a.reflections));
  VKR_TRY(make_tex(out, 0, VKR_FMT_RGBA8_SRGB, "defered_shading.out", &a.out));
  load_mat(a.inverse_camera, consts->inverse_camera);
  a.pr.tg = tanf(consts->fovy / 2.0f);
  a.pr.aspect = consts->aspect; a.pr.znear = consts->znear; a.pr.zfar = consts->zfar;
  a.min_roughness = push->min_max_roughness[0];
  a.max_roughness = push->min_max_roughness[1];
  a.show_ao = push->show_ao;
  const bool fast = same_layout(a.normal, a.albedo) && same_layout(a.normal, a.material) && same_layout(a.normal, a.depth0) && a.normal.w >= 2 &&
                    a.depth1.w >= 2 && a.occlusion.w >= 2 && a.reflections.w >= 2 && a.brdf.w >= 2 && !(switches() & VKR_SWITCH_SHADING_GENERIC);
  dim3 block(64, 4);
  if (fast) hipLaunchKernelGGL(k_defered_shading<true>, grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_defered_shading<false>, grid2d(a.out.w, a.out.h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("defered_shading");
}
