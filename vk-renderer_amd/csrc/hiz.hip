// hiz.hip — Hi-Z build: programs "downsample_gbuffer" and "depth_mips".
//
// Reference: src/downsample_pass.cpp:25-143, shaders/advanced_ssr/downsample_gbuffer.frag:12-37,
// shaders/advanced_ssr/depth_mips.frag:7-15.  Depth stays in its stored D24 integer form:
// min() of quantised values re-quantises to itself, so the path is bit-exact integer work.
// Roofline: HBM.  D1 moves 15 B per full-res pixel, D2 1.667 B.
#include "vkr_host.hpp"
#include <algorithm>

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
    uint2 r0 = *(const uint2*)(d0.p + toff(d0, px, py, 4));
    uint2 r1 = *(const uint2*)(d0.p + toff(d0, px, (py + 1), 4));
    da = r0.x & 0xFFFFFFu; db = r0.y & 0xFFFFFFu; dc = r1.x & 0xFFFFFFu; dd = r1.y & 0xFFFFFFu;
    uint2 m0 = *(const uint2*)(n0.p + toff(n0, px, py, 4));
    uint2 m1 = *(const uint2*)(n0.p + toff(n0, px, (py + 1), 4));
    na = m0.x; nb = m0.y; nc = m1.x; nd = m1.y;
    uint2 w0 = *(const uint2*)(v0.p + toff(v0, px, py, 4));
    uint2 w1 = *(const uint2*)(v0.p + toff(v0, px, (py + 1), 4));
    va = w0.x; vb = w0.y; vc = w1.x; vd = w1.y;
  } else {  // texelFetch out of bounds -> 0
    auto ld = [&](const Tex& t, int lx, int ly) -> uint32_t {
      return (lx < t.w && ly < t.h) ? *(const uint32_t*)(t.p + toff(t, lx, ly, 4)) : 0u;
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

// depth_mips.frag: mip i = 2x2 min of mip i-1.  The reference records one full-screen draw per mip
// (downsample_pass.cpp:107-129: L-2 dependent passes, 10 at 4K).  Here one workgroup reduces a
// 32x32 block of the source mip through LDS and writes up to FUSED_LEVELS consecutive mips, so
// the whole chain is two launches.  A destination texel (X,Y) of level k exists iff X < W_k and
// Y < H_k (W_k = max(1, W >> k)); a non-existent texel is only ever read when its parent extent
// is 1, where the reference's out-of-bounds texelFetch returns 0 — so non-existent texels are
// carried as 0 through LDS and never stored.
#define FUSED_LEVELS 5
struct MipChainArgs {
  Tex src;
  Tex dst[FUSED_LEVELS];
  int levels;
};

VKR_DEV uint32_t min4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return min(min(a, b), min(c, d)); }

__global__ __launch_bounds__(256) void k_depth_mips_fused(MipChainArgs a) {
  __shared__ uint32_t s_l1[16][16], s_l2[8][8], s_l3[4][4], s_l4[2][2];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  {  // level 1: one thread per texel, 2x2 quad of the source as two 8-byte row loads
    const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
    const Tex& src = a.src;
    uint32_t v = 0u;
    if (x < a.dst[0].w && y < a.dst[0].h) {
      uint32_t q0, q1, q2, q3;
      if (2 * x + 1 < src.w && 2 * y + 1 < src.h) {
        uint2 r0 = *(const uint2*)(src.p + toff(src, (2 * x), (2 * y), 4));
        uint2 r1 = *(const uint2*)(src.p + toff(src, (2 * x), (2 * y + 1), 4));
        q0 = r0.x & 0xFFFFFFu; q1 = r0.y & 0xFFFFFFu; q2 = r1.x & 0xFFFFFFu; q3 = r1.y & 0xFFFFFFu;
      } else {  // parent extent 1: texelFetch out of bounds -> 0
        auto ld = [&](int lx, int ly) -> uint32_t {
          return (lx < src.w && ly < src.h) ? (*(const uint32_t*)(src.p + toff(src, lx, ly, 4)) & 0xFFFFFFu) : 0u;
        };
        q0 = ld(2 * x, 2 * y); q1 = ld(2 * x + 1, 2 * y); q2 = ld(2 * x, 2 * y + 1); q3 = ld(2 * x + 1, 2 * y + 1);
      }
      v = min4(q0, q1, q2, q3);
      *texel_ptr<uint32_t>(a.dst[0], x, y) = v;
    }
    s_l1[ty][tx] = v;
  }
  if (a.levels < 2) return;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lx = threadIdx.x & 7, ly = threadIdx.x >> 3;
    const int x = blockIdx.x * 8 + lx, y = blockIdx.y * 8 + ly;
    uint32_t v = 0u;
    if (x < a.dst[1].w && y < a.dst[1].h) {
      v = min4(s_l1[2 * ly][2 * lx], s_l1[2 * ly][2 * lx + 1], s_l1[2 * ly + 1][2 * lx], s_l1[2 * ly + 1][2 * lx + 1]);
      *texel_ptr<uint32_t>(a.dst[1], x, y) = v;
    }
    s_l2[ly][lx] = v;
  }
  if (a.levels < 3) return;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int lx = threadIdx.x & 3, ly = threadIdx.x >> 2;
    const int x = blockIdx.x * 4 + lx, y = blockIdx.y * 4 + ly;
    uint32_t v = 0u;
    if (x < a.dst[2].w && y < a.dst[2].h) {
      v = min4(s_l2[2 * ly][2 * lx], s_l2[2 * ly][2 * lx + 1], s_l2[2 * ly + 1][2 * lx], s_l2[2 * ly + 1][2 * lx + 1]);
      *texel_ptr<uint32_t>(a.dst[2], x, y) = v;
    }
    s_l3[ly][lx] = v;
  }
  if (a.levels < 4) return;
  __syncthreads();
  if (threadIdx.x < 4) {
    const int lx = threadIdx.x & 1, ly = threadIdx.x >> 1;
    const int x = blockIdx.x * 2 + lx, y = blockIdx.y * 2 + ly;
    uint32_t v = 0u;
    if (x < a.dst[3].w && y < a.dst[3].h) {
      v = min4(s_l3[2 * ly][2 * lx], s_l3[2 * ly][2 * lx + 1], s_l3[2 * ly + 1][2 * lx], s_l3[2 * ly + 1][2 * lx + 1]);
      *texel_ptr<uint32_t>(a.dst[3], x, y) = v;
    }
    s_l4[ly][lx] = v;
  }
  if (a.levels < 5) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int x = blockIdx.x, y = blockIdx.y;
    if (x < a.dst[4].w && y < a.dst[4].h)
      *texel_ptr<uint32_t>(a.dst[4], x, y) = min4(s_l4[0][0], s_l4[0][1], s_l4[1][0], s_l4[1][1]);
  }
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
  if (depth->mip_count > VKR_MAX_MIPS) { set_error("depth_mips: bad mip count"); return VKR_ERR_MIPS; }
  for (uint32_t s = src_mip; s + 1 < depth->mip_count; s += FUSED_LEVELS) {
    MipChainArgs a;
    VKR_TRY(make_tex(depth, (int)s, VKR_FMT_D24_UNORM_S8, "depth_mips.src", &a.src));
    if (a.src.pitch % 8 != 0 || (uintptr_t)a.src.p % 8 != 0) { set_error("depth_mips: rows must be 8-byte aligned"); return VKR_ERR_LAYOUT; }
    a.levels = (int)std::min<uint32_t>(FUSED_LEVELS, depth->mip_count - 1 - s);
    for (int k = 0; k < FUSED_LEVELS; k++)
      VKR_TRY(make_tex(depth, (int)s + 1 + (k < a.levels ? k : a.levels - 1), VKR_FMT_D24_UNORM_S8, "depth_mips.dst", &a.dst[k]));
    // one workgroup per 32x32 source block; level-1 extent decides the grid
    dim3 grid((a.dst[0].w + 15) / 16, (a.dst[0].h + 15) / 16);
    hipLaunchKernelGGL(k_depth_mips_fused, grid, dim3(256), 0, (hipStream_t)stream, a);
    VKR_TRY(launch_status("depth_mips"));
  }
  return VKR_OK;
}
