// hit_exchange.hip — request / reply for the hit colours of the stochastic SSR (multi-GPU; SURVEY.md 8(e) (2), no
// reference counterpart: the reference is single-GPU).  shaders/advanced_ssr/filter.comp:112-134 reads the albedo
// bilinearly at the hit position of every valid ray — anywhere in the frame.  Instead of all-gathering the albedo of the
// whole frame into every rank (464 MB per rank and frame at 15360x8640 on 8 GPUs), a rank asks the owners for exactly the
// footprint rows it does not hold:
//
//   vkr_hit_requests   per valid ray of the rank's window: the two texel rows of the bilinear footprint of
//                      texture(albedo, hit uv) — the very rows vkr_device.hpp sample_srgb_rgb() touches — that lie outside
//                      the window, binned by owning rank.  Pass 1 (out == NULL) counts per owner, pass 2 writes the
//                      requests segment by segment.  A request is {frame row, left texel of the pair}: 8 bytes.
//   vkr_hit_reply      the owner reads the two texels of every request it received from its own albedo: 8 bytes back.
//   vkr_hit_scatter    the requester writes the replies into its whole-frame albedo image at their frame positions; the
//                      filter kernel is unchanged — it finds the texels where the all-gather would have put them.
//
// In the benchmark frame 6 % of the rays have a footprint row on another rank.  Byte moving and integer work, HBM-bound.
#include "vkr_host.hpp"

namespace vkr {

#define HIT_MAX_WORLD 16

struct HitReqArgs {
  Tex rays;                 // RGBA16_UNORM, half-res window
  int aw, ah;               // albedo frame extent (full-res)
  uint32_t bounds[HIT_MAX_WORLD + 1];  // strip r owns frame rows [bounds[r], bounds[r + 1])
  uint32_t world;
  uint32_t win0, win1;      // the rows this rank holds: [win0, win1)
  uint32_t* counts;         // pass 1: per owner
  uint32_t* cursors;        // pass 2: per owner, zeroed by the caller
  uint32_t seg[HIT_MAX_WORLD];  // pass 2: first request of owner o's segment
  vkr_hit_request* out;     // NULL: pass 1
};

// One thread per ray texel of the window.  Requests of a block are reserved per owner with one global atomic.
__global__ __launch_bounds__(256) void k_hit_requests(HitReqArgs a) {
  __shared__ uint32_t s_n[HIT_MAX_WORLD], s_base[HIT_MAX_WORLD];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  if (tid < HIT_MAX_WORLD) s_n[tid] = 0u;
  __syncthreads();
  const int lx = blockIdx.x * 64 + threadIdx.x, ly = blockIdx.y * 4 + threadIdx.y;
  // up to two rows per ray
  uint32_t row[2], x = 0u, owner[2], slot[2];
  int n = 0;
  bool have = false;
  f2 uv = mk2(0.0f, 0.0f);
  if (lx < a.rays.w && ly < a.rays.h && blockIdx.y != gridDim.y - 1) {
    const uint2 v = *(const uint2*)(a.rays.p + toff(a.rays, lx, ly, 8));
    if ((v.y >> 16) != 0xFFFFu) { have = true; uv = mk2(unorm16_to_float(v.x & 0xFFFFu), unorm16_to_float(v.x >> 16)); }  // filter.comp:93-95: w != 1
  }
  // the grid has one block row more than the window: its first thread stands for the ray texels OUTSIDE the frame that the
  // filter's apron reads at the frame's left / right edge — they read 0, i.e. uv (0, 0) with w = 0 != 1: a hit
  if (blockIdx.y == gridDim.y - 1 && blockIdx.x == 0 && tid == 0) { have = true; uv = mk2(0.0f, 0.0f); }
  if (have) {
    // texture(albedo, uv): vkr_device.hpp bilinear_taps_u32 — texel rows clamp(y0, y0 + 1), pair (xs, xs + 1)
    const float fx = cfma(uv.x, (float)a.aw, -0.5f), fy = cfma(uv.y, (float)a.ah, -0.5f);
    const int x0 = f2i(floorf(fx)), y0 = f2i(floorf(fy));
    x = (uint32_t)iclamp(x0, 0, a.aw - 2);
    const uint32_t r0 = (uint32_t)iclamp(y0, 0, a.ah - 1), r1 = (uint32_t)iclamp(y0 + 1, 0, a.ah - 1);
    if (r0 < a.win0 || r0 >= a.win1) row[n++] = r0;
    if (r1 != r0 && (r1 < a.win0 || r1 >= a.win1)) row[n++] = r1;
  }
  for (int k = 0; k < n; k++) {
    uint32_t o = 0;
    while (o + 1 < a.world && row[k] >= a.bounds[o + 1]) ++o;
    owner[k] = o;
    slot[k] = atomicAdd(&s_n[o], 1u);
  }
  __syncthreads();
  if (tid < (int)a.world && s_n[tid]) {
    if (a.out) s_base[tid] = a.seg[tid] + atomicAdd(&a.cursors[tid], s_n[tid]);
    else atomicAdd(&a.counts[tid], s_n[tid]);
  }
  if (!a.out) return;
  __syncthreads();
  for (int k = 0; k < n; k++) {
    vkr_hit_request r;
    r.row = row[k]; r.x = x;
    a.out[s_base[owner[k]] + slot[k]] = r;
  }
}

// two texels (8 bytes) per request, from the owner's window image
__global__ __launch_bounds__(256) void k_hit_reply(Tex albedo, const vkr_hit_request* req, uint32_t count, uint64_t* replies, uint32_t* errors) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const vkr_hit_request r = req[i];
  const int ly = (int)r.row - albedo.oy, lx = (int)r.x - albedo.ox;
  if (ly < 0 || ly >= albedo.h || lx < 0 || lx + 1 >= albedo.w) { atomicAdd(errors, 1u); replies[i] = 0ull; return; }  // not mine: the row bounds of the ranks disagree
  const U32x2 t = load_u32x2(albedo.p + toff(albedo, lx, ly, 4));
  replies[i] = (uint64_t)t.x | ((uint64_t)t.y << 32);
}

__global__ __launch_bounds__(256) void k_hit_scatter(Tex frame_albedo, const vkr_hit_request* req, const uint64_t* replies, uint32_t count) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const vkr_hit_request r = req[i];
  const uint64_t v = replies[i];
  uint32_t* dst = (uint32_t*)(const_cast<uint8_t*>(frame_albedo.p) + toff(frame_albedo, (int)r.x, (int)r.row, 4));
  dst[0] = (uint32_t)v; dst[1] = (uint32_t)(v >> 32);
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_hit_requests(const vkr_img* rays, uint32_t albedo_width, uint32_t albedo_height, const uint32_t* row_bounds, uint32_t world,
                                uint32_t window_row0, uint32_t window_row1, uint32_t* counts, uint32_t* cursors, const uint32_t* segments,
                                vkr_hit_request* out, void* stream) {
  if (!row_bounds || (!out && !counts) || (out && (!cursors || !segments))) { set_error("hit_requests: NULL argument"); return VKR_ERR_NULL; }
  if (world < 1 || world > HIT_MAX_WORLD) { set_error("hit_requests: world %u (1..%d)", world, HIT_MAX_WORLD); return VKR_ERR_EXTENT; }
  if (albedo_width < 2 || albedo_height < 1 || row_bounds[0] != 0 || row_bounds[world] != albedo_height || window_row0 >= window_row1 || window_row1 > albedo_height) {
    set_error("hit_requests: frame %ux%u, window rows [%u, %u), bounds [%u .. %u]", albedo_width, albedo_height, window_row0, window_row1, row_bounds[0], row_bounds[world]);
    return VKR_ERR_EXTENT;
  }
  HitReqArgs a;
  VKR_TRY(make_tex(rays, 0, VKR_FMT_RGBA16_UNORM, "hit_requests.rays", &a.rays));
  a.aw = (int)albedo_width; a.ah = (int)albedo_height;
  for (uint32_t r = 0; r <= world; r++) {
    if (r && row_bounds[r] <= row_bounds[r - 1]) { set_error("hit_requests: row bounds must increase"); return VKR_ERR_EXTENT; }
    a.bounds[r] = row_bounds[r];
  }
  for (uint32_t r = world + 1; r <= HIT_MAX_WORLD; r++) a.bounds[r] = albedo_height;
  a.world = world; a.win0 = window_row0; a.win1 = window_row1;
  a.counts = counts; a.cursors = cursors; a.out = out;
  for (uint32_t r = 0; r < HIT_MAX_WORLD; r++) a.seg[r] = (out && r < world) ? segments[r] : 0u;
  const dim3 block(64, 4);
  dim3 grid = grid2d(a.rays.w, a.rays.h, block);
  grid.y += 1;  // the extra row: see the kernel
  hipLaunchKernelGGL(k_hit_requests, grid, block, 0, (hipStream_t)stream, a);
  return launch_status("hit_requests");
}

extern "C" int vkr_hit_reply(const vkr_img* albedo, const vkr_hit_request* requests, uint32_t count, uint64_t* replies, uint32_t* error_counter, void* stream) {
  if (count == 0) return VKR_OK;
  if (!requests || !replies || !error_counter) { set_error("hit_reply: NULL argument"); return VKR_ERR_NULL; }
  Tex t;
  VKR_TRY(make_tex(albedo, 0, VKR_FMT_RGBA8_SRGB, "hit_reply.albedo", &t));
  hipLaunchKernelGGL(k_hit_reply, dim3((count + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, t, requests, count, replies, error_counter);
  return launch_status("hit_reply");
}

extern "C" int vkr_hit_scatter(const vkr_img* frame_albedo, const vkr_hit_request* requests, const uint64_t* replies, uint32_t count, void* stream) {
  if (count == 0) return VKR_OK;
  if (!requests || !replies) { set_error("hit_scatter: NULL argument"); return VKR_ERR_NULL; }
  Tex t;
  VKR_TRY(make_tex(frame_albedo, 0, VKR_FMT_RGBA8_SRGB, "hit_scatter.frame_albedo", &t));
  if (t.ox != 0 || t.oy != 0 || t.w != t.fw || t.h != t.fh) { set_error("hit_scatter: the destination must be the whole-frame image"); return VKR_ERR_EXTENT; }
  hipLaunchKernelGGL(k_hit_scatter, dim3((count + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, t, requests, replies, count);
  return launch_status("hit_scatter");
}
