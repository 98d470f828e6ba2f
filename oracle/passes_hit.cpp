// ORACLE — TEST INFRASTRUCTURE ONLY (see glsl.hpp header).  Parity: UNPINNED (no reference counterpart: the reference is
// single-GPU; these restate csrc/hit_exchange.hip so that the request / reply exchange of the tiled frame has a CPU twin
// for the gloo tests and a per-kernel checker on the GPU).
//
// passes_hit.cpp — hit colours / hit normals by request / reply (include/vkr_postfx.h): which footprint rows of
// texture(albedo, hit uv) (filter.comp:112-134) and texture(normal, hit uv) (trace.comp:103-109) a rank lacks, the owner's
// answer, and where the answer goes.  Sequential, so request order is deterministic (the product's order is not: compare
// as sets per owner).
#include <vector>

#include "shader_common.hpp"

using namespace oracle;

namespace {
struct Req { uint32_t code, owner; };

void footprint(std::vector<Req>& out, float u, float v, int w, int h, uint32_t lo, uint32_t hi, uint32_t tag, uint32_t shift,
               const uint32_t* bounds, uint32_t world) {
  const int x0 = f2i(floorf(cfma(u, (float)w, -0.5f))), y0 = f2i(floorf(cfma(v, (float)h, -0.5f)));
  const uint32_t x = (uint32_t)clamp(x0, 0, w - 2);
  const uint32_t r0 = (uint32_t)clamp(y0, 0, h - 1), r1 = (uint32_t)clamp(y0 + 1, 0, h - 1);
  const bool want0 = r0 < lo || r0 >= hi, want1 = r1 != r0 && (r1 < lo || r1 >= hi);
  auto owner = [&](uint32_t r) { uint32_t o = 0; while (o + 1 < world && (r << shift) >= bounds[o + 1]) ++o; return o; };
  const uint32_t o0 = owner(r0), o1 = owner(r1);
  if (want0 && want1 && o0 == o1) { out.push_back({r0 | (x << 14) | VKR_HIT_BOTH_ROWS | tag, o0}); return; }
  if (want0) out.push_back({r0 | (x << 14) | tag, o0});
  if (want1) out.push_back({r1 | (x << 14) | tag, o1});
}
}  // namespace

// (the product's fifth argument is a workspace its pass 1 leaves for its pass 2: no use here)
extern "C" int vkr_ref_hit_requests(const vkr_hit_sources* src, const uint32_t* row_bounds, uint32_t world, uint32_t* counts, uint32_t* /*workspace*/,
                                    const uint32_t* segments, vkr_hit_request* out) {
  std::vector<uint32_t> cursors(world, 0u);
  Image RAYS(*src->rays);
  std::vector<Req> reqs;
  for (int ly = 0; ly < RAYS.h(); ly++)
    for (int lx = 0; lx < RAYS.w(); lx++) {
      uint16_t t[4];
      std::memcpy(t, RAYS.texel_ptr(lx, ly, 0), 8);
      if (t[3] != 0xFFFFu)  // filter.comp:93-95: w != 1
        footprint(reqs, unorm16_to_float(t[0]), unorm16_to_float(t[1]), (int)src->albedo_width, (int)src->albedo_height, src->window_row0,
                  src->window_row1, 0u, 0u, row_bounds, world);
      if (src->pending_mask && *Image(*src->pending_mask).texel_ptr(lx, ly, 0) != 0) {
        const float* pd = (const float*)Image(*src->pending_data).texel_ptr(2 * lx, ly, 0);
        footprint(reqs, pd[4], pd[5], (int)src->normal_width, (int)src->normal_height, src->normal_row0, src->normal_row1, VKR_HIT_NORMAL, 1u,
                  row_bounds, world);
      }
    }
  // the ray texels outside the frame that the filter's apron reads: 0, i.e. uv (0, 0), w = 0: a hit
  footprint(reqs, 0.0f, 0.0f, (int)src->albedo_width, (int)src->albedo_height, src->window_row0, src->window_row1, 0u, 0u, row_bounds, world);
  for (const Req& r : reqs) {
    if (out) out[segments[r.owner] + cursors[r.owner]++] = r.code;
    else counts[r.owner]++;
  }
  return 0;
}

extern "C" int vkr_ref_hit_reply(const vkr_img* albedo, const vkr_img* normals, const vkr_hit_request* requests, uint32_t count, void* replies,
                                 uint32_t* error_counter) {
  uint32_t* out = (uint32_t*)replies;
  for (uint32_t i = 0; i < count; i++) {
    const uint32_t r = requests[i];
    const bool nrm = (r & VKR_HIT_NORMAL) != 0u, both = (r & VKR_HIT_BOTH_ROWS) != 0u;
    if (nrm && !normals) { (*error_counter)++; std::memset(out + 4 * i, 0, 16); continue; }
    Image T(nrm ? *normals : *albedo);
    const int ly = (int)(r & 0x3FFFu) - T.oy(), lx = (int)((r >> 14) & 0x3FFFu) - T.ox();
    if (ly < 0 || ly + (both ? 1 : 0) >= T.h() || lx < 0 || lx + 1 >= T.w()) { (*error_counter)++; std::memset(out + 4 * i, 0, 16); continue; }
    out[4 * i + 0] = T.load_u32(lx, ly, 0); out[4 * i + 1] = T.load_u32(lx + 1, ly, 0);
    out[4 * i + 2] = T.load_u32(lx, ly + (both ? 1 : 0), 0); out[4 * i + 3] = T.load_u32(lx + 1, ly + (both ? 1 : 0), 0);
  }
  return 0;
}

extern "C" int vkr_ref_hit_scatter(const vkr_img* frame_albedo, const vkr_img* frame_normals, const vkr_hit_request* requests, const void* replies,
                                   uint32_t count) {
  const uint32_t* in = (const uint32_t*)replies;
  for (uint32_t i = 0; i < count; i++) {
    const uint32_t r = requests[i];
    Image T((r & VKR_HIT_NORMAL) ? *frame_normals : *frame_albedo);
    const int row = (int)(r & 0x3FFFu), x = (int)((r >> 14) & 0x3FFFu);
    T.store_u32(x, row, 0, in[4 * i + 0]); T.store_u32(x + 1, row, 0, in[4 * i + 1]);
    if (r & VKR_HIT_BOTH_ROWS) { T.store_u32(x, row + 1, 0, in[4 * i + 2]); T.store_u32(x + 1, row + 1, 0, in[4 * i + 3]); }
  }
  return 0;
}
