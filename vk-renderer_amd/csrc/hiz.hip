// hiz.hip — Hi-Z build: programs "downsample_gbuffer" and "depth_mips".
//
// Reference: src/downsample_pass.cpp:25-143, shaders/advanced_ssr/downsample_gbuffer.frag:12-37,
// shaders/advanced_ssr/depth_mips.frag:7-15.  Depth stays in its stored D24 integer form:
// min() of quantised values re-quantises to itself, so the path is bit-exact integer work.
// Roofline: HBM.  D1 moves 15 B per full-res pixel, D2 1.667 B.
#include "vkr_host.hpp"

namespace vkr {

// One thread per half-res pixel.  Each lane reads its 2x2 quad as two 8-byte row loads
// (lanes contiguous -> 512 B per wave per row), for depth, normal and velocity alike, and
// selects in registers; the three outputs are 4-byte coalesced stores.
__global__ __launch_bounds__(256) void k_downsample_gbuffer(Tex d0, Tex d1, Tex n0, Tex v0, Tex on, Tex ov) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= on.w || y >= on.h) return;
  const int px = 2 * x, py = 2 * y;
  const bool has_x1 = px + 1 < d0.w, has_y1 = py + 1 < d0.h;  // only false for odd extents
  uint32_t da, db, dc, dd, na, nb, nc, nd, va, vb, vc, vd;
  if (has_x1 && has_y1) {
    uint2 r0 = *(const uint2*)(d0.p + (size_t)py * d0.pitch + (size_t)px * 4);
    uint2 r1 = *(const uint2*)(d0.p + (size_t)(py + 1) * d0.pitch + (size_t)px * 4);
    da = r0.x & 0xFFFFFFu; db = r0.y & 0xFFFFFFu; dc = r1.x & 0xFFFFFFu; dd = r1.y & 0xFFFFFFu;
    uint2 m0 = *(const uint2*)(n0.p + (size_t)py * n0.pitch + (size_t)px * 4);
    uint2 m1 = *(const uint2*)(n0.p + (size_t)(py + 1) * n0.pitch + (size_t)px * 4);
    na = m0.x; nb = m0.y; nc = m1.x; nd = m1.y;
    uint2 w0 = *(const uint2*)(v0.p + (size_t)py * v0.pitch + (size_t)px * 4);
    uint2 w1 = *(const uint2*)(v0.p + (size_t)(py + 1) * v0.pitch + (size_t)px * 4);
    va = w0.x; vb = w0.y; vc = w1.x; vd = w1.y;
  } else {  // texelFetch out of bounds -> 0
    auto ld = [&](const Tex& t, int lx, int ly) -> uint32_t {
      return (lx < t.w && ly < t.h) ? *(const uint32_t*)(t.p + (size_t)ly * t.pitch + (size_t)lx * 4) : 0u;
    };
    da = ld(d0, px, py) & 0xFFFFFFu; db = ld(d0, px + 1, py) & 0xFFFFFFu;
    dc = ld(d0, px, py + 1) & 0xFFFFFFu; dd = ld(d0, px + 1, py + 1) & 0xFFFFFFu;
    na = ld(n0, px, py); nb = ld(n0, px + 1, py); nc = ld(n0, px, py + 1); nd = ld(n0, px + 1, py + 1);
    va = ld(v0, px, py); vb = ld(v0, px + 1, py); vc = ld(v0, px, py + 1); vd = ld(v0, px + 1, py + 1);
  }
  const uint32_t mn = min(min(da, db), min(dc, dd));
  // downsample_gbuffer.frag:24-32: first match among d1, d2, d3, else (0,0)
  uint32_t n = na, v = va;
  if (mn == db) { n = nb; v = vb; }
  else if (mn == dc) { n = nc; v = vc; }
  else if (mn == dd) { n = nd; v = vd; }
  *texel_ptr<uint32_t>(on, x, y) = n;
  *texel_ptr<uint32_t>(ov, x, y) = v;
  *texel_ptr<uint32_t>(d1, x, y) = mn;
}

// mip i = 2x2 min of mip i-1 (depth_mips.frag).  One thread per destination texel.
__global__ __launch_bounds__(256) void k_depth_mip(Tex src, Tex dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dst.w || y >= dst.h) return;
  auto ld = [&](int lx, int ly) -> uint32_t {
    return (lx < src.w && ly < src.h) ? (*(const uint32_t*)(src.p + (size_t)ly * src.pitch + (size_t)lx * 4) & 0xFFFFFFu) : 0u;
  };
  uint32_t a, b, c, d;
  if (2 * x + 1 < src.w && 2 * y + 1 < src.h) {
    uint2 r0 = *(const uint2*)(src.p + (size_t)(2 * y) * src.pitch + (size_t)(2 * x) * 4);
    uint2 r1 = *(const uint2*)(src.p + (size_t)(2 * y + 1) * src.pitch + (size_t)(2 * x) * 4);
    a = r0.x & 0xFFFFFFu; b = r0.y & 0xFFFFFFu; c = r1.x & 0xFFFFFFu; d = r1.y & 0xFFFFFFu;
  } else {
    a = ld(2 * x, 2 * y); b = ld(2 * x + 1, 2 * y); c = ld(2 * x, 2 * y + 1); d = ld(2 * x + 1, 2 * y + 1);
  }
  *texel_ptr<uint32_t>(dst, x, y) = min(min(a, b), min(c, d));
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_downsample_gbuffer(const vkr_img* depth, const vkr_img* normal, const vkr_img* velocity,
                                      const vkr_img* out_normal, const vkr_img* out_velocity, void* stream) {
  if (!depth || depth->mip_count < 2) {  // downsample_pass.cpp:37-39
    set_error("downsample_gbuffer: Can't downsample depth texture with 1 mip level");
    return VKR_ERR_MIPS;
  }
  Tex d0, d1, n0, v0, on, ov;
  VKR_TRY(make_tex(depth, 0, VKR_FMT_D24_UNORM_S8, "downsample_gbuffer.depth", &d0));
  VKR_TRY(make_tex(depth, 1, VKR_FMT_D24_UNORM_S8, "downsample_gbuffer.depth mip1", &d1));
  VKR_TRY(make_tex(normal, 0, VKR_FMT_RG16_UNORM, "downsample_gbuffer.normal", &n0));
  VKR_TRY(make_tex(velocity, 0, VKR_FMT_RG16_SFLOAT, "downsample_gbuffer.velocity", &v0));
  VKR_TRY(make_tex(out_normal, 0, VKR_FMT_RG16_UNORM, "downsample_gbuffer.out_normal", &on));
  VKR_TRY(make_tex(out_velocity, 0, VKR_FMT_RG16_SFLOAT, "downsample_gbuffer.out_velocity", &ov));
  if (!same_window(d1, on) || !same_window(on, ov)) {  // downsample_pass.cpp:48-50
    set_error("downsample_gbuffer: Output textures have different sizes");
    return VKR_ERR_EXTENT;
  }
  if (!same_window(d0, n0) || !same_window(d0, v0)) {
    set_error("downsample_gbuffer: source textures have different sizes");
    return VKR_ERR_EXTENT;
  }
  if ((d0.pitch | n0.pitch | v0.pitch) % 8 != 0 || ((uintptr_t)d0.p | (uintptr_t)n0.p | (uintptr_t)v0.p) % 8 != 0) {
    set_error("downsample_gbuffer: source rows must be 8-byte aligned");
    return VKR_ERR_LAYOUT;
  }
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_downsample_gbuffer, grid2d(on.w, on.h, block), block, 0, (hipStream_t)stream, d0, d1, n0, v0, on, ov);
  return launch_status("downsample_gbuffer");
}

extern "C" int vkr_depth_mips(const vkr_img* depth, uint32_t src_mip, void* stream) {
  if (!depth) { set_error("depth_mips: NULL image"); return VKR_ERR_NULL; }
  for (uint32_t i = src_mip + 1; i < depth->mip_count; i++) {
    Tex src, dst;
    VKR_TRY(make_tex(depth, (int)i - 1, VKR_FMT_D24_UNORM_S8, "depth_mips.src", &src));
    VKR_TRY(make_tex(depth, (int)i, VKR_FMT_D24_UNORM_S8, "depth_mips.dst", &dst));
    if (src.pitch % 8 != 0 || (uintptr_t)src.p % 8 != 0) { set_error("depth_mips: rows must be 8-byte aligned"); return VKR_ERR_LAYOUT; }
    dim3 block(64, 4);
    hipLaunchKernelGGL(k_depth_mip, grid2d(dst.w, dst.h, block), block, 0, (hipStream_t)stream, src, dst);
    VKR_TRY(launch_status("depth_mips"));
  }
  return VKR_OK;
}
