// ssr_indirect.hip — the tile-classified stochastic SSR trace the reference ships but leaves commented
// out of AdvancedSSR::run (advanced_ssr.cpp:548-550): programs
// "sssr_classification" (classification.comp), "sssr_trace_indirect" (trace_indirect.comp) and the
// SSSR_Clear transfer (advanced_ssr.cpp:440-452).
//
// HIP has no dispatch-indirect: the trace is launched over the upper bound of the tile list and
// every workgroup compares its index with the count the classification left in the indirect-args
// buffer (VkDispatchIndirectCommand.x), exiting at once when it is beyond — an empty workgroup costs
// a few cycles, and nothing returns to the host between the two passes.
#include "vkr_host.hpp"
#include "hiz_march.hpp"
#include "ssr_sampling.hpp"

namespace vkr {

// advanced_ssr.cpp:447-450: VkDispatchIndirectCommand{0, 1, 1} into both argument buffers
__global__ void k_sssr_clear_indirect(uint32_t* reflective, uint32_t* glossy) {
  if (threadIdx.x < 3) {
    const uint32_t v = threadIdx.x == 0 ? 0u : 1u;
    reflective[threadIdx.x] = v;
    glossy[threadIdx.x] = v;
  }
}

// classification.comp:38-98.  One wave per 8x8 tile; the shared-memory tree (offsets 32,16,...,1,
// element t += element t+offset) is the same pairing as wave shuffles by the same offsets, so lane 0
// ends with the identical fp32 sum.
__global__ __launch_bounds__(64) void k_sssr_classification(Tex material, int32_t* reflective_tiles, int32_t* glossy_tiles,
                                                            uint32_t* reflective_args, uint32_t* glossy_args, int tex_w, int tex_h,
                                                            float max_roughness, float glossy_value) {
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  const int tid = threadIdx.x;
  srgb_lut_stage(s_lut, tid, 64);
  __syncthreads();
  const int px = blockIdx.x * 8 + (tid & 7), py = blockIdx.y * 8 + (tid >> 3);
  float sampled_roughness = 1.0f;
  if (px < tex_w && py < tex_h) {
    const f2 screen_uv = mk2((float)px / (float)tex_w, (float)py / (float)tex_h);
    sampled_roughness = sample_srgb_channel(material, screen_uv, 1, s_lut);
  }
  float v = mixf(0.0f, max_roughness, sampled_roughness);
#pragma unroll
  for (int offset = 32; offset != 0; offset >>= 1) v = v + __shfl_down(v, offset, 64);
  if (tid == 0) {
    const float average_roughness = v / 64.0f;
    const int tiles_x = (tex_w + 7) / 8;
    const int tile_index = blockIdx.y * tiles_x + blockIdx.x;
    if (average_roughness < glossy_value) reflective_tiles[atomicAdd(reflective_args, 1u)] = tile_index;
    else glossy_tiles[atomicAdd(glossy_args, 1u)] = tile_index;
  }
}

struct TraceIndirectArgs {
  Pyramid depth;
  Tex normal, material, out_ray;
  const float4* halton;
  const int32_t* tiles;
  const uint32_t* args;  // VkDispatchIndirectCommand: x = number of tiles
  Mat4 normal_mat;
  Proj pr;
  uint32_t frame_random, reflection_type;
  float max_roughness;
};

// trace_indirect.comp:43-135.  One wave per listed tile.
__global__ __launch_bounds__(64) void k_sssr_trace_indirect(TraceIndirectArgs a) {
  if (blockIdx.x >= *a.args) return;  // beyond the indirect count
  __shared__ uint4 s_mip[16];
  __shared__ float s_lut[VKR_SRGB_LUT_SIZE];
  const int tid = threadIdx.x;
  srgb_lut_stage(s_lut, tid, 64);
  if (tid < 16) s_mip[tid] = mip_descriptor(a.depth.mip[tid < a.depth.count ? tid : 0]);
  __syncthreads();
  const int tile_index = a.tiles[blockIdx.x];
  const int tile_width = (a.out_ray.fw + 7) / 8;
  const int gx = 8 * (tile_index % tile_width) + (tid & 7), gy = 8 * (tile_index / tile_width) + (tid >> 3);
  if (gx >= a.out_ray.fw || gy >= a.out_ray.fh) return;
  const f2 tex_size = mk2((float)a.out_ray.fw, (float)a.out_ray.fh);
  const f2 screen_uv = mk2((float)gx / tex_size.x, (float)gy / tex_size.y);  // no +0.5 (trace_indirect.comp:51)
  const Proj pr = a.pr;
  const Tex& depth0 = a.depth.mip[0];

  float roughness = sample_srgb_channel(a.material, screen_uv, 1, s_lut);
  roughness = mixf(0.0f, a.max_roughness, roughness);
  roughness *= roughness;
  const float pixel_depth = sample<FmtD24>(depth0, screen_uv);
  const f3 pnw = decode_normal(sample<FmtRG16U>(a.normal, screen_uv));
  const f3 pixel_normal = normalize(xyz(mul(a.normal_mat, mk4(pnw.x, pnw.y, pnw.z, 0.0f))));
  const f3 view_vec = reconstruct_view_vec(screen_uv, pixel_depth, pr);

  const float rdot = dot(screen_uv, mk2(12.9898f, 78.233f));
  const float rnd01 = fractf(sin_hash_arg(rdot) * 43758.5453f);  // sin in double: it picks the Halton entry
  const uint32_t index = (f2u(rnd01 * (float)VKR_HALTON_SEQ_SIZE) + a.frame_random) & (VKR_HALTON_SEQ_SIZE - 1);
  const float4 hv = a.halton[index];

  f3 tangent = get_tangent(pixel_normal);
  const f3 bitangent = normalize(cross(pixel_normal, tangent));
  tangent = normalize(cross(bitangent, pixel_normal));
  f3 view_dir = -normalize(view_vec);
  view_dir = mk3(dot(view_dir, tangent), dot(view_dir, bitangent), dot(view_dir, pixel_normal));
  const f3 brdf_norm = sampleGGXVNDF(view_dir, roughness, roughness, hv.x, hv.z, hv.w);
  const f3 N = (brdf_norm.x * tangent + brdf_norm.y * bitangent) + brdf_norm.z * pixel_normal;
  const f3 R = reflect(view_vec, N);

  f3 ray_start = project_view_vec(view_vec + 0.001f * pixel_normal, pr);
  ray_start.z -= 0.0001f;
  f3 ray_dir = project_view_vec(view_vec + R, pr) - ray_start;
  ray_dir = ray_dir * ((1.0f - ray_start.z) / ray_dir.z);

  // mirror tiles: hierarchical_raymarch(DEPTH, start, dir, 0, 50); glossy: (.., 1, 25)  (:98-102)
  const int min_mip = a.reflection_type == 0 ? 0 : 1;
  const int max_steps = a.reflection_type == 0 ? 50 : 25;
  MarchEnv env;
  env.mip_table = s_mip;
  env.mip_count = a.depth.count;
  env.screen_size = mk2((float)depth0.fw, (float)depth0.fh);
  env.screen_size_inv = mk2(1.0f / env.screen_size.x, 1.0f / env.screen_size.y);
  const float uvo = 0.005f * __builtin_ldexpf(1.0f, min_mip);
  env.uv_offset_abs = mk2(uvo / env.screen_size.x, uvo / env.screen_size.y);
  env.pr = pr;
  env.horizon_d2 = 0.0f;
  env.min_mip = min_mip;
  RayConst rc;
  rc.origin = ray_start;
  rc.direction = ray_dir;
  rc.inv_direction = safe_inverse(ray_dir);
  rc.normal = pixel_normal;
  rc.view_vec = view_vec;
  RayState st;
  st.t = initial_advance(env, rc);
  st.h = 0.0f; st.mip = min_mip; st.i = 0;
  bool more = max_steps > 0;
#pragma unroll 1
  while (more) more = march_step<false, 0>(env, rc, st, max_steps);
  const f3 out_ray = madd(rc.origin, st.t, rc.direction);

  bool valid_hit = true;  // i <= max always (screen_trace.glsl:97)
  {
    const f2 ray_step = mk2(fabsf(out_ray.x - ray_start.x) * tex_size.x, fabsf(out_ray.y - ray_start.y) * tex_size.y);
    if (vmax(ray_step.x, ray_step.y) < 2.0f) valid_hit = false;
  }
  if (valid_hit) {
    const f3 hnw = decode_normal(sample<FmtRG16U>(a.normal, xy(out_ray)));
    const f3 hit_normal = xyz(mul(a.normal_mat, mk4(hnw.x, hnw.y, hnw.z, 0.0f)));
    if (dot(hit_normal, R) > 0.0f || dot(pixel_normal, R) < 0.0f) valid_hit = false;
  }
  if (valid_hit && a.reflection_type == 0) {
    const float hit_z = linearize_depth2_unorm(sample<FmtD24>(depth0, xy(out_ray)), pr.znear, pr.zfar);
    const float ray_z = linearize_depth2(out_ray.z, pr.znear, pr.zfar);
    if (ray_z > hit_z + 0.3f || ray_z < hit_z - 0.1f) valid_hit = false;
  }
  const int lx = gx - a.out_ray.ox, ly = gy - a.out_ray.oy;
  if (lx < 0 || ly < 0 || lx >= a.out_ray.w || ly >= a.out_ray.h) return;
  uint2 o;
  o.x = float_to_unorm16(out_ray.x) | (float_to_unorm16(out_ray.y) << 16);
  o.y = float_to_unorm16(out_ray.z) | (float_to_unorm16(valid_hit ? pixel_depth : 1.0f) << 16);
  *texel_ptr<uint2>(a.out_ray, lx, ly) = o;
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_sssr_clear_indirect(uint32_t* reflective_args, uint32_t* glossy_args, void* stream) {
  if (!reflective_args || !glossy_args) { set_error("sssr_clear: NULL argument buffer"); return VKR_ERR_NULL; }
  hipLaunchKernelGGL(k_sssr_clear_indirect, dim3(1), dim3(64), 0, (hipStream_t)stream, reflective_args, glossy_args);
  return launch_status("sssr_clear");
}

extern "C" int vkr_sssr_classification(const vkr_img* material, int32_t* reflective_tiles, int32_t* glossy_tiles,
                                       uint32_t* reflective_args, uint32_t* glossy_args,
                                       const vkr_classification_push* push, void* stream) {
  if (!push || !reflective_tiles || !glossy_tiles || !reflective_args || !glossy_args) { set_error("sssr_classification: NULL argument"); return VKR_ERR_NULL; }
  if (push->width <= 0 || push->height <= 0) { set_error("sssr_classification: empty extent"); return VKR_ERR_EXTENT; }
  Tex m;
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "sssr_classification.material", &m));
  dim3 grid((push->width + 7) / 8, (push->height + 7) / 8);
  hipLaunchKernelGGL(k_sssr_classification, grid, dim3(64), 0, (hipStream_t)stream, m, reflective_tiles, glossy_tiles, reflective_args,
                     glossy_args, push->width, push->height, push->max_roughness, push->glossy_value);
  return launch_status("sssr_classification");
}

extern "C" int vkr_sssr_trace_indirect(const vkr_img* depth, const vkr_img* normal, const vkr_img* material,
                                       const vkr_trace_params* params, const float* halton_vec4, const vkr_img* out_rays,
                                       const int32_t* tiles, const uint32_t* indirect_args, uint32_t max_tiles,
                                       const vkr_trace_indirect_push* push, void* stream) {
  if (!params || !push || !halton_vec4 || !tiles || !indirect_args || !depth) { set_error("sssr_trace_indirect: NULL argument"); return VKR_ERR_NULL; }
  if (push->reflection_type > 1) { set_error("sssr_trace_indirect: reflection_type %u", push->reflection_type); return VKR_ERR_EXTENT; }
  TraceIndirectArgs a;
  if (depth->mip_count < 1 || depth->mip_count > VKR_MAX_MIPS) { set_error("sssr_trace_indirect: bad depth mip count"); return VKR_ERR_MIPS; }
  a.depth.count = (int)depth->mip_count;
  for (int i = 0; i < a.depth.count; i++) {
    VKR_TRY(make_tex(depth, i, VKR_FMT_D24_UNORM_S8, "sssr_trace_indirect.depth", &a.depth.mip[i]));
    const Tex& m = a.depth.mip[i];
    if (m.ox != 0 || m.oy != 0 || m.w != m.fw || m.h != m.fh || m.w > 65535 || m.h > 65535) {
      set_error("sssr_trace_indirect: the depth pyramid must cover the whole frame");
      return VKR_ERR_EXTENT;
    }
  }
  for (int i = a.depth.count; i < 16; i++) a.depth.mip[i] = a.depth.mip[0];
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "sssr_trace_indirect.normal", &a.normal));
  VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "sssr_trace_indirect.material", &a.material));
  VKR_TRY(make_tex(out_rays, 0, VKR_FMT_RGBA16_UNORM, "sssr_trace_indirect.rays", &a.out_ray));
  a.halton = (const float4*)halton_vec4;
  a.tiles = tiles;
  a.args = indirect_args;
  load_mat(a.normal_mat, params->normal_mat);
  a.pr.tg = tanf(params->fovy / 2.0f);
  a.pr.aspect = params->aspect; a.pr.znear = params->znear; a.pr.zfar = params->zfar;
  a.frame_random = params->frame_random;
  a.reflection_type = push->reflection_type;
  a.max_roughness = push->max_roughness;
  const uint32_t all_tiles = (uint32_t)((a.out_ray.fw + 7) / 8) * (uint32_t)((a.out_ray.fh + 7) / 8);
  const uint32_t bound = max_tiles < all_tiles ? max_tiles : all_tiles;
  if (bound == 0) return VKR_OK;
  hipLaunchKernelGGL(k_sssr_trace_indirect, dim3(bound), dim3(64), 0, (hipStream_t)stream, a);
  return launch_status("sssr_trace_indirect");
}
