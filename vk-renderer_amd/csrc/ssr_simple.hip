// ssr_simple.hip — program "ssr": the single-bounce mirror SSR of src/ssr.cpp:10-73 +
// shaders/ssr/shader.frag:27-102 (full-resolution fragment pass; not called by the reference's
// frame loop).  Depth is read through a NEAREST sampler with U/W
// clamp-to-border (ssr.cpp:21-28); the march is the generic hierarchical_raymarch
// (screen_trace.glsl:51-100): most detailed mip 0 of the full-res depth, at most 100 steps.
// Bound: dependent texel fetches of the march (21.3 B per pixel compulsory).
#include "vkr_host.hpp"
#include "hiz_march.hpp"

namespace vkr {

struct SsrArgs {
  Pyramid depth;  // all mips of the full-res depth image
  Tex normal, frame, material, out;
  Mat4 camera_normal;
  Proj pr;
  uint32_t frame_format;  // VKR_FMT_RGBA8_SRGB or VKR_FMT_RGBA8_UNORM
};

// texture() through the NEAREST / clamp-to-border(U) / clamp-to-edge(V) depth sampler
VKR_DEV float sample_depth_nearest(const Tex& d, f2 uv) {
  const int x = f2i(floorf(uv.x * (float)d.fw));
  const int y = iclamp(f2i(floorf(uv.y * (float)d.fh)), 0, d.fh - 1);
  if (x < 0 || x >= d.fw) return 0.0f;  // opaque-black border
  return fetch<FmtD24>(d, x, y);
}
VKR_DEV float smoothstep1(float e0, float e1, float x) {
  const float t = vclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return (t * t) * (3.0f - 2.0f * t);
}
// brdf.glsl:31-38
VKR_DEV float DistributionGGX(f3 N, f3 H, float alpha) {
  const float NoH = dot(N, H);
  const float alpha2 = alpha * alpha;
  const float NoH2 = NoH * NoH;
  const float den = NoH2 * alpha2 + (1.0f - NoH2);
  return (((NoH2 > 0.0f) ? 1.0f : 0.0f) * alpha2) / ((VKR_PI * den) * den);
}
// rgba of texture(frame_tex, uv)
VKR_DEV f4 sample_frame(const Tex& t, f2 uv, bool srgb, const float* lut) {
  const BilinearTaps b = bilinear_taps_u32(t, uv);
  auto dec = [&](uint32_t v) {
    if (srgb) return mk4(lut[v & 0xFFu], lut[(v >> 8) & 0xFFu], lut[(v >> 16) & 0xFFu], unorm8_to_float(v >> 24));
    return mk4(unorm8_to_float(v & 0xFFu), unorm8_to_float((v >> 8) & 0xFFu), unorm8_to_float((v >> 16) & 0xFFu), unorm8_to_float(v >> 24));
  };
  return mix4(mix4(dec(b.t00), dec(b.t10), b.fx), mix4(dec(b.t01), dec(b.t11), b.fx), b.fy);
}

__global__ __launch_bounds__(256) void k_ssr_simple(SsrArgs a) {
  __shared__ uint4 s_mip[16];
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  const int tid = threadIdx.x;
  srgb_lut_stage(s_lut, tid, 256);
  if (tid < 16) s_mip[tid] = mip_descriptor(a.depth.mip[tid < a.depth.count ? tid : 0]);
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int lx = (blockIdx.x * 4 + wave) * 8 + (lane & 7);
  const int ly = blockIdx.y * 8 + (lane >> 3);
  if (lx >= a.out.w || ly >= a.out.h) return;
  const int gx = a.out.ox + lx, gy = a.out.oy + ly;
  const Proj pr = a.pr;
  const Tex& depth0 = a.depth.mip[0];
  f4 out_reflection = mk4(0, 0, 0, 0);
  do {
    const f2 screen_uv = mk2(((float)gx + 0.5f) / (float)a.out.fw, ((float)gy + 0.5f) / (float)a.out.fh);
    const f2 tex_size = mk2((float)a.frame.fw, (float)a.frame.fh);
    const f2 aligned_screen_uv = mk2(floorf(screen_uv.x * tex_size.x) / tex_size.x + 0.5f / tex_size.x,
                                     floorf(screen_uv.y * tex_size.y) / tex_size.y + 0.5f / tex_size.y);
    const float roughness = sample_srgb_channel(a.material, screen_uv, 1, s_lut);
    const float pixel_depth = sample_depth_nearest(depth0, aligned_screen_uv);
    const f3 pnw = decode_normal(sample<FmtRG16U>(a.normal, aligned_screen_uv));
    const f3 pixel_normal = normalize(xyz(mul(a.camera_normal, mk4(pnw.x, pnw.y, pnw.z, 0.0f))));
    // depth may be the border value 0 here -> general linearize, not the [0,1]-only fast path? 0 is in range.
    const f3 view_vec = reconstruct_view_vec(aligned_screen_uv, pixel_depth, pr);
    const f3 R = reflect(view_vec, pixel_normal);
    const f3 start = project_view_vec(view_vec + 0.0005f * pixel_normal, pr);
    const f3 p = project_view_vec(view_vec + R, pr);
    const f3 delta = normalize(p - start);
    if (fabsf(delta.z) < 0.0000001f) break;
    float t_bound = (1.0f - start.z) / delta.z;
    const float u_bound = vmax((1.0f - start.x) / delta.x, -start.x / delta.x);
    const float v_bound = vmax((1.0f - start.y) / delta.y, -start.y / delta.y);
    t_bound = vmin(t_bound, vmin(u_bound, v_bound));
    const f3 end = start + t_bound * delta;

    MarchEnv env;
    env.mip_table = s_mip;
    env.mip_count = a.depth.count;
    env.screen_size = mk2((float)depth0.fw, (float)depth0.fh);
    env.screen_size_inv = mk2(1.0f / env.screen_size.x, 1.0f / env.screen_size.y);
    env.uv_offset_abs = mk2(0.005f / env.screen_size.x, 0.005f / env.screen_size.y);
    env.pr = pr;
    env.horizon_d2 = 0.0f;
    RayConst rc;
    rc.origin = start;
    rc.direction = end - start;
    rc.inv_direction = safe_inverse(rc.direction);
    rc.normal = pixel_normal;
    rc.view_vec = view_vec;
    RayState st;
    st.t = initial_advance(env, rc);
    st.h = 0.0f; st.mip = 0; st.i = 0;
    bool more = true;
#pragma unroll 1
    while (more) more = march_step<false, 0>(env, rc, st, 100);
    const f3 out_ray = madd(rc.origin, st.t, rc.direction);  // valid_hit = (i <= 100) is always true

    const f2 dist0 = mk2(fabsf(out_ray.x - start.x), fabsf(out_ray.y - start.y));
    if (dist0.x < 2.0f / tex_size.x && dist0.y < 2.0f / tex_size.y) break;
    const f3 hnw = decode_normal(sample<FmtRG16U>(a.normal, xy(out_ray)));
    const f3 hit_normal = xyz(mul(a.camera_normal, mk4(hnw.x, hnw.y, hnw.z, 0.0f)));
    if (dot(hit_normal, R) > 0.0f) break;
    const float hit_depth = sample_depth_nearest(depth0, xy(out_ray));
    if (out_ray.z > hit_depth + 0.0001f) break;
    const f2 fov = mk2(0.05f * (tex_size.y / tex_size.x), 0.05f * 1.0f);
    const float bx = smoothstep1(0.0f, fov.x, out_ray.x) * (1.0f - smoothstep1(1.0f - fov.x, 1.0f, out_ray.x));
    const float by = smoothstep1(0.0f, fov.y, out_ray.y) * (1.0f - smoothstep1(1.0f - fov.y, 1.0f, out_ray.y));
    const float coef = bx * by;
    const f4 c = sample_frame(a.frame, xy(out_ray), a.frame_format == VKR_FMT_RGBA8_SRGB, s_lut);
    const float k = DistributionGGX(pixel_normal, pixel_normal, roughness);
    const float ndr = vmax(dot(pixel_normal, R), 0.0f);
    out_reflection = mk4(((coef * c.x) * k) * ndr, ((coef * c.y) * k) * ndr, ((coef * c.z) * k) * ndr, ((coef * c.w) * k) * ndr);
  } while (false);
  *texel_ptr<uint32_t>(a.out, lx, ly) = float_to_unorm8(out_reflection.x) | (float_to_unorm8(out_reflection.y) << 8) |
                                        (float_to_unorm8(out_reflection.z) << 16) | (float_to_unorm8(out_reflection.w) << 24);
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_ssr(const vkr_img* normal, const vkr_img* depth, const vkr_img* frame, const vkr_ssr_params* params,
                       const vkr_img* material, const vkr_img* out, void* stream) {
  if (!params || !depth || !frame) { set_error("ssr: NULL argument"); return VKR_ERR_NULL; }
  SsrArgs a;
  if (depth->mip_count < 1 || depth->mip_count > VKR_MAX_MIPS) { set_error("ssr: bad depth mip count"); return VKR_ERR_MIPS; }
  a.depth.count = (int)depth->mip_count;
  for (int i = 0; i < a.depth.count; i++) {
    VKR_TRY(make_tex(depth, i, VKR_FMT_D24_UNORM_S8, "ssr.depth", &a.depth.mip[i]));
    const Tex& m = a.depth.mip[i];
    if (m.ox != 0 || m.oy != 0 || m.w != m.fw || m.h != m.fh || m.w > 65535 || m.h > 65535) {
      set_error("ssr: the depth pyramid must cover the whole frame");
      return VKR_ERR_EXTENT;
    }
  }
  for (int i = a.depth.count; i < 16; i++) a.depth.mip[i] = a.depth.mip[0];
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "ssr.normal", &a.normal));
  a.frame_format = frame->format;
  if (frame->format != VKR_FMT_RGBA8_SRGB && frame->format != VKR_FMT_RGBA8_UNORM) { set_error("ssr: frame must be RGBA8 (sRGB or UNORM)"); return VKR_ERR_FORMAT; }
  VKR_TRY(make_tex(frame, 0, frame->format, "ssr.frame", &a.frame));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "ssr.material", &a.material));
  VKR_TRY(make_tex(out, 0, VKR_FMT_RGBA8_UNORM, "ssr.out", &a.out));
  load_mat(a.camera_normal, params->normal_mat);
  a.pr.tg = tanf(params->fovy / 2.0f);
  a.pr.aspect = params->aspect; a.pr.znear = params->znear; a.pr.zfar = params->zfar;
  dim3 grid((a.out.w + 31) / 32, (a.out.h + 7) / 8);
  hipLaunchKernelGGL(k_ssr_simple, grid, dim3(256), 0, (hipStream_t)stream, a);
  return launch_status("ssr");
}
