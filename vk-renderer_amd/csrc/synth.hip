// synth.hip — synthetic G-buffer generator (replaces the raster stage
// scene_renderer.cpp:140-220 + gbuf/opaque_taa.{vert,frag}) and the
// streaming-read microbenchmark that provides the measured HBM roofline denominator.
//
// Scene (frozen): ground plane y = 0; back wall z = 12, |x| <= 12, y in [0,7]; 6 x 4 spheres
// of radius 0.6 at (-5+2i, 0.6, 3+2j); everything else is sky (depth 1).  Outputs use the
// reference's attachment formats (scene_renderer.cpp:13-43).
#include "vkr_host.hpp"

namespace vkr {

VKR_DEV uint32_t pcg(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
VKR_DEV uint32_t hash3(uint32_t a, uint32_t b, uint32_t seed) { return pcg(seed ^ pcg(a ^ pcg(b))); }
VKR_DEV float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

struct SynthArgs {
  Tex depth, normal, albedo, material, velocity;
  Mat4 c2w, prev_mvp, mvp;
  Proj pr;
  uint32_t seed, depth_only, textured_roughness;
};

__global__ __launch_bounds__(256) void k_synth_gbuffer(SynthArgs a) {
  const int lx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ly = blockIdx.y * blockDim.y + threadIdx.y;
  if (lx >= a.depth.w || ly >= a.depth.h) return;
  const int gx = a.depth.ox + lx, gy = a.depth.oy + ly;
  const float znear = a.pr.znear, zfar = a.pr.zfar;
  const f3 eye = xyz(mul(a.c2w, mk4(0, 0, 0, 1)));
  const float u = ((float)gx + 0.5f) / (float)a.depth.fw, v = ((float)gy + 0.5f) / (float)a.depth.fh;
  const float xd = 2.0f * u - 1.0f, yd = 2.0f * v - 1.0f;
  const f3 dir = xyz(mul(a.c2w, mk4((xd * a.pr.aspect) * a.pr.tg, yd * a.pr.tg, -1.0f, 0.0f)));

  float best_t = zfar;
  uint32_t id = 0xFFFFFFFFu;
  f3 nrm = mk3(0, 0, -1);
  if (dir.y < 0.0f) {  // ground
    float t = -eye.y / dir.y;
    if (t > znear && t < best_t) { best_t = t; id = 0u; nrm = mk3(0, 1, 0); }
  }
  if (dir.z > 0.0f) {  // back wall
    float t = (12.0f - eye.z) / dir.z;
    if (t > znear && t < best_t) {
      float hx = eye.x + t * dir.x, hy = eye.y + t * dir.y;
      if (fabsf(hx) <= 12.0f && hy >= 0.0f && hy <= 7.0f) { best_t = t; id = 1u; nrm = mk3(0, 0, -1); }
    }
  }
  const float SPHERE_R = 0.6f;
  const float qa = dot(dir, dir);
  for (int j = 0; j < 4; j++) {
    for (int i = 0; i < 6; i++) {
      const f3 c = mk3(-5.0f + 2.0f * (float)i, SPHERE_R, 3.0f + 2.0f * (float)j);
      const f3 oc = eye - c;
      const float b = dot(dir, oc);
      const float c0 = dot(oc, oc) - SPHERE_R * SPHERE_R;
      const float disc = b * b - qa * c0;
      if (disc > 0.0f) {
        const float t = (-b - sqrtf(disc)) / qa;
        if (t > znear && t < best_t) {
          best_t = t;
          id = 2u + (uint32_t)(j * 6 + i);
          nrm = normalize((eye + t * dir) - c);
        }
      }
    }
  }

  uint32_t d24, nt = 0, at = 0, mt = 0, vt = 0;
  if (id == 0xFFFFFFFFu) {  // sky
    d24 = 0xFFFFFFu;
    const f2 en = encode_normal(mk3(0, 0, -1));
    nt = float_to_unorm16(en.x) | (float_to_unorm16(en.y) << 16);
    at = float_to_srgb8(0.45f) | (float_to_srgb8(0.65f) << 8) | (float_to_srgb8(0.9f) << 16) | (float_to_unorm8(1.0f) << 24);
    mt = float_to_srgb8(0.5f) | (float_to_srgb8(1.0f) << 8) | (float_to_srgb8(0.0f) << 16) | (float_to_unorm8(0.5f) << 24);
    vt = 0u;
  } else {
    const float z_view = -best_t;
    const float dz = encode_depth(z_view, znear, zfar);
    d24 = (uint32_t)rintf(vclamp(dz, 0.0f, 1.0f) * 16777215.0f) & 0xFFFFFFu;
    if (!a.depth_only) {
      const f3 P = eye + best_t * dir;
      const f3 base = mk3(0.3f + 0.7f * u01(hash3(id, 1u, a.seed)), 0.3f + 0.7f * u01(hash3(id, 2u, a.seed)),
                          0.3f + 0.7f * u01(hash3(id, 3u, a.seed)));
      float roughness = 0.1f + 0.8f * u01(hash3(id, 4u, a.seed));
      if (a.textured_roughness) {  // VKR_SYNTH_TEXTURED_ROUGHNESS: per-texel roughness
        const float n = u01(hash3((uint32_t)gx, (uint32_t)gy, a.seed ^ 0x7E57u)) - 0.5f;
        roughness = roughness + 0.3f * n;
        roughness = roughness < 0.02f ? 0.02f : (roughness > 1.0f ? 1.0f : roughness);
      }
      const float metallic = (hash3(id, 5u, a.seed) & 1u) ? 1.0f : 0.0f;
      const f2 en = encode_normal(nrm);
      int cx, cy;
      if (id == 0u) { cx = f2i(floorf(P.x)); cy = f2i(floorf(P.z)); }
      else if (id == 1u) { cx = f2i(floorf(P.x)); cy = f2i(floorf(P.y)); }
      else { cx = f2i(floorf(8.0f * en.x)); cy = f2i(floorf(8.0f * en.y)); }
      const float checker = (hash3((uint32_t)cx, (uint32_t)cy, a.seed ^ id) & 1u) ? 1.0f : 0.5f;
      nt = float_to_unorm16(en.x) | (float_to_unorm16(en.y) << 16);
      at = float_to_srgb8(base.x * checker) | (float_to_srgb8(base.y * checker) << 8) | (float_to_srgb8(base.z * checker) << 16) |
           (float_to_unorm8(1.0f) << 24);
      mt = float_to_srgb8(0.5f) | (float_to_srgb8(roughness) << 8) | (float_to_srgb8(metallic) << 16) | (float_to_unorm8(0.5f) << 24);
      const f4 cp = mul(a.prev_mvp, mk4(P.x, P.y, P.z, 1.0f)), cc = mul(a.mvp, mk4(P.x, P.y, P.z, 1.0f));
      const float vx = 0.5f * (cp.x / cp.w - cc.x / cc.w), vy = 0.5f * (cp.y / cp.w - cc.y / cc.w);
      vt = float_to_half_bits(vx) | (float_to_half_bits(vy) << 16);
    }
  }
  *texel_ptr<uint32_t>(a.depth, lx, ly) = d24;
  if (!a.depth_only) {
    *texel_ptr<uint32_t>(a.normal, lx, ly) = nt;
    *texel_ptr<uint32_t>(a.albedo, lx, ly) = at;
    *texel_ptr<uint32_t>(a.material, lx, ly) = mt;
    *texel_ptr<uint32_t>(a.velocity, lx, ly) = vt;
  }
}

// float4 streaming read: every lane sums 16-byte loads of a grid-strided range; one float per
// block leaves the chip so the loads cannot be elided.
__global__ __launch_bounds__(256) void k_stream_read(const float4* __restrict__ src, uint64_t n_vec, float* sink, uint32_t sink_len) {
  float acc = 0.0f;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    acc += ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w)) + ((v2.x + v2.y) + (v2.z + v2.w)) + ((v3.x + v3.y) + (v3.z + v3.w));
  }
  for (; i < n_vec; i += stride) { float4 v = src[i]; acc += (v.x + v.y) + (v.z + v.w); }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x % sink_len] = (part[0] + part[1]) + (part[2] + part[3]);
}

// Self-test (tests only): counts disagreements between the cheap exact forms and their definitions.
//  [0] div_normal(nf, d(f-n)-f) vs IEEE '/' over all 2^24 stored depths, (n,f) = (znear, zfar)
//  [1] div_normal(a, b) vs a / b on a hashed set of normal-range operands
//  [2..4] d24 / unorm16 / unorm8 decode vs the correctly rounded quotient
__global__ __launch_bounds__(256) void k_selftest_division(uint32_t* counters, float znear, float zfar) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^24-1
  const float d = d24_to_float(i);
  const float ref = (znear * zfar) / cfma(d, zfar - znear, -zfar);  // the contract's linearize_depth2 with the IEEE quotient
  if (linearize_depth2_unorm(d, znear, zfar) != ref) atomicAdd(&counters[0], 1u);
  const float a = __uint_as_float(0x30000000u + (pcg(i) & 0x1FFFFFFFu));        // ~4.7e-10 .. 1.8e+19
  float b = __uint_as_float(0x30000000u + (pcg(i ^ 0x9E3779B9u) & 0x1FFFFFFFu));
  if (i & 1u) b = -b;
  if (div_normal(a, b) != a / b) atomicAdd(&counters[1], 1u);
  if (d != (float)i / 16777215.0f) atomicAdd(&counters[2], 1u);
  if (i < 65536u && unorm16_to_float(i) != (float)i / 65535.0f) atomicAdd(&counters[3], 1u);
  if (i < 256u && unorm8_to_float(i) != (float)i / 255.0f) atomicAdd(&counters[4], 1u);
}

// [0] div_normal(g + 0.5, size) vs the IEEE quotient for every pixel centre g < size of every extent size <= max_size:
// the screen_uv of a pass that computes it with div_normal
__global__ __launch_bounds__(256) void k_selftest_pixel_uv(uint32_t* counter, uint32_t max_size) {
  const uint32_t size = blockIdx.y + 1u, g = blockIdx.x * blockDim.x + threadIdx.x;
  if (size > max_size || g >= size) return;
  const float a = (float)g + 0.5f, b = (float)size;
  if (div_normal(a, b) != a / b) atomicAdd(counter, 1u);
}

// counters[0]: floats x in [2^-96, FLT_MAX] (ALL of them: bit patterns 0x0F800000 .. 0x7F7FFFFF) whose sqrt_ieee(x) differs
// from sqrtf(x); counters[1]: the same for normalize()'s reciprocal, rcp_ieee_normal(s) vs 1.0f / s over every float s in
// [2^-48, 2^64]
__global__ __launch_bounds__(256) void k_selftest_sqrt(uint32_t* counters) {
  const uint64_t first = 0x0F800000ull, last = 0x7F7FFFFFull;
  uint32_t bad = 0u, bad_rcp = 0u;
  for (uint64_t b = first + (uint64_t)blockIdx.x * 256u + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * 256u) {
    const float x = __uint_as_float((uint32_t)b);
    if (sqrt_ieee(x) != sqrtf(x)) ++bad;
    if (x >= 0x1p-48f && x <= 0x1p64f && rcp_ieee_normal(x) != 1.0f / x) ++bad_rcp;
  }
  if (bad) atomicAdd(&counters[0], bad);
  if (bad_rcp) atomicAdd(&counters[1], bad_rcp);
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_selftest_sqrt(uint32_t* device_counters2, void* stream) {
  if (!device_counters2) { set_error("selftest_sqrt: NULL counters"); return VKR_ERR_NULL; }
  hipLaunchKernelGGL(k_selftest_sqrt, dim3(8192), dim3(256), 0, (hipStream_t)stream, device_counters2);
  return launch_status("selftest_sqrt");
}

extern "C" int vkr_selftest_pixel_uv(uint32_t* device_counter, uint32_t max_size, void* stream) {
  if (!device_counter || max_size == 0 || max_size > 65535u) { set_error("selftest_pixel_uv: bad arguments"); return VKR_ERR_NULL; }
  hipLaunchKernelGGL(k_selftest_pixel_uv, dim3((max_size + 255u) / 256u, max_size), dim3(256), 0, (hipStream_t)stream, device_counter, max_size);
  return launch_status("selftest_pixel_uv");
}

extern "C" int vkr_selftest_division(uint32_t* device_counters5, float znear, float zfar, void* stream) {
  if (!device_counters5) { set_error("selftest_division: NULL counters"); return VKR_ERR_NULL; }
  hipLaunchKernelGGL(k_selftest_division, dim3((1u << 24) / 256), dim3(256), 0, (hipStream_t)stream, device_counters5, znear, zfar);
  return launch_status("selftest_division");
}

extern "C" int vkr_synth_gbuffer(const vkr_img* depth, const vkr_img* normal, const vkr_img* albedo,
                                 const vkr_img* material, const vkr_img* velocity, const vkr_synth_params* params,
                                 void* stream) {
  if (!params) { set_error("synth_gbuffer: NULL params"); return VKR_ERR_NULL; }
  SynthArgs a;
  a.depth_only = (params->flags & VKR_SYNTH_DEPTH_ONLY) ? 1u : 0u;
  a.textured_roughness = (params->flags & VKR_SYNTH_TEXTURED_ROUGHNESS) ? 1u : 0u;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "synth_gbuffer.depth", &a.depth));
  if (!a.depth_only) {
    VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "synth_gbuffer.normal", &a.normal));
    VKR_TRY(make_tex(albedo, 0, VKR_FMT_RGBA8_SRGB, "synth_gbuffer.albedo", &a.albedo));
    VKR_TRY(make_tex(material, 0, VKR_FMT_RGBA8_SRGB, "synth_gbuffer.material", &a.material));
    VKR_TRY(make_tex(velocity, 0, VKR_FMT_RG16_SFLOAT, "synth_gbuffer.velocity", &a.velocity));
    if (!same_window(a.depth, a.normal) || !same_window(a.depth, a.albedo) || !same_window(a.depth, a.material) ||
        !same_window(a.depth, a.velocity)) {
      set_error("synth_gbuffer: attachments differ in extent");
      return VKR_ERR_EXTENT;
    }
  } else {
    a.normal = a.albedo = a.material = a.velocity = a.depth;
  }
  load_mat(a.c2w, params->camera_to_world);
  load_mat(a.prev_mvp, params->prev_mvp);
  load_mat(a.mvp, params->mvp);
  a.pr.tg = tanf(params->fovy / 2.0f);
  a.pr.aspect = params->aspect; a.pr.znear = params->znear; a.pr.zfar = params->zfar;
  a.seed = params->seed;
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_synth_gbuffer, grid2d(a.depth.w, a.depth.h, block), block, 0, (hipStream_t)stream, a);
  return launch_status("synth_gbuffer");
}

extern "C" int vkr_stream_read(const void* src, uint64_t bytes, float* sink, uint32_t sink_len, void* stream) {
  if (!src || !sink || sink_len == 0 || ((uintptr_t)src % 16) != 0) { set_error("stream_read: bad arguments"); return VKR_ERR_NULL; }
  const uint64_t n_vec = bytes / 16;
  hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const float4*)src, n_vec, sink, sink_len);
  return launch_status("stream_read");
}
