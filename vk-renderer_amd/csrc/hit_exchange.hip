// hit_exchange.hip — request / reply for the hit colours of the stochastic SSR (multi-GPU; SURVEY.md 8(e) (2), no
// reference counterpart: the reference is single-GPU).  shaders/advanced_ssr/filter.comp:112-134 reads the albedo
// bilinearly at the hit position of every valid ray — anywhere in the frame.  Instead of all-gathering the albedo of the
// whole frame into every rank (464 MB per rank and frame at 15360x8640 on 8 GPUs), a rank asks the owners for exactly the
// footprint rows it does not hold:
//
//   vkr_hit_requests   per valid ray of the rank's window: the two texel rows of the bilinear footprint of
//                      texture(albedo, hit uv) — the very rows vkr_device.hpp sample_srgb_rgb() touches — that lie outside
//                      the window, binned by owning rank.  Pass 1 (out == NULL) counts per owner, pass 2 writes the
//                      requests segment by segment.  A request is 4 bytes: frame row, left texel of the pair, "and the row
//                      below" (both rows of a footprint usually belong to one owner), which surface.
//   vkr_hit_reply      the owner reads the texel pair(s) of every request it received from its own window: 16 bytes back.
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
  uint32_t bounds[HIT_MAX_WORLD + 1];  // strip r owns full-res frame rows [bounds[r], bounds[r + 1])
  uint32_t world;
  uint32_t win0, win1;      // the full-res rows this rank holds: [win0, win1)
  // normals (pending rays of the windowed trace); has_normals == 0: none
  uint32_t has_normals;
  Tex pend_mask, pend_data;
  int nw, nh;               // downsampled-normal frame extent (half-res)
  uint32_t nrow0, nrow1;    // the half-res rows held
  uint32_t* counts;         // pass 1: per owner
  uint32_t* workspace;      // [block][HIT_MAX_WORLD]: what each block of pass 1 counted per owner; pass 2 turns it into the blocks' bases
  uint32_t seg[HIT_MAX_WORLD];  // pass 2: first request of owner o's segment
  uint32_t cap[HIT_MAX_WORLD];  // pass 2: room in owner o's segment (requests beyond it are dropped: vkr_hit_requests_bounded)
  vkr_hit_request* out;     // NULL: pass 1
  uint32_t* dropped;        // bounded pass 2: set to 1 when a request did not fit its segment (or NULL)
};

// One thread per ray texel of a 64 x 16 tile.  A launch has at most HIT_BLOCKS blocks of 1024 threads; a block walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... of the window, HIT_BATCH of them at a time with all their ray loads in flight
// together (one dependent load per tile and wave would leave the kernel waiting on memory latency).  Pass 1 adds the block's
// requests per owner to the rank's counters with one atomic per owner and block — all counters of a rank share a cache
// line, a device-scope atomic on one line costs ~11 ns whoever issues it, and one reservation per 64 x 4 tile (16 k tiles of
// a 7680 x 540 strip) made this byte-moving kernel take 0.16 ms — and leaves what it counted in the workspace.  Pass 2 has
// the same blocks walk the same tiles: a block's requests for owner o start at segment[o] + what the blocks before it
// counted, so it needs neither a second count nor a global atomic; within a block's range the slots go out in a fixed order (tile,
// wave, request of the ray, lane), so the same rays always put the same bytes on the wire.
#define HIT_BLOCKS 256
static_assert(HIT_BLOCKS * HIT_MAX_WORLD == VKR_HIT_WORKSPACE_WORDS, "the workspace holds one count per block and owner");
#define HIT_BATCH 4
#define HIT_TILE_H 16
struct HitTile { int tiles_x, tiles_y; };  // tiles of the window plus one apron row of tiles (see hit_emit)
struct HitRay { uint2 v; uint32_t pending; int lx, ly; bool inside, apron; float2 hit_uv; };  // hit_uv: of a pending ray (pending data, texel 1)

VKR_DEV HitRay hit_load(const HitReqArgs& a, const HitTile& g, int tile, int tid) {
  HitRay r;
  const int bx = tile % g.tiles_x, by = tile / g.tiles_x;
  r.lx = bx * 64 + (tid & 63); r.ly = by * HIT_TILE_H + (tid >> 6);
  const bool apron_row = by == g.tiles_y - 1;
  r.inside = tile < g.tiles_x * g.tiles_y && r.lx < a.rays.w && r.ly < a.rays.h && !apron_row;
  // the last row of tiles lies below the window: its first thread stands for the ray texels OUTSIDE the frame that the
  // filter's apron reads at the frame's left / right edge — they read 0, i.e. uv (0, 0) with w = 0 != 1: a hit
  r.apron = tile < g.tiles_x * g.tiles_y && apron_row && bx == 0 && tid == 0;
  r.v = make_uint2(0u, 0xFFFF0000u); r.pending = 0u;
  if (r.inside) {
    r.v = *(const uint2*)(a.rays.p + toff(a.rays, r.lx, r.ly, 8));
    if (a.has_normals) r.pending = *texel_ptr<uint8_t>(a.pend_mask, r.lx, r.ly);
  }
  return r;
}

// per ray and surface one request (both footprint rows from one owner) or two (the rows belong to different strips)
VKR_DEV int hit_emit(const HitReqArgs& a, const HitRay& r, uint32_t* code, uint32_t* owner) {
  int n = 0;
  bool have = false;
  f2 uv = mk2(0.0f, 0.0f);
  if (r.inside && (r.v.y >> 16) != 0xFFFFu) { have = true; uv = mk2(unorm16_to_float(r.v.x & 0xFFFFu), unorm16_to_float(r.v.x >> 16)); }  // filter.comp:93-95: w != 1
  if (r.apron) have = true;
  // rows r0 <= r1 <= r0 + 1 of a footprint, [lo, hi) the rows held, shift: strips are cut at even full-res rows
  auto emit = [&](uint32_t r0, uint32_t r1, uint32_t x, uint32_t lo, uint32_t hi, uint32_t tag, uint32_t shift) {
    const bool want0 = r0 < lo || r0 >= hi, want1 = r1 != r0 && (r1 < lo || r1 >= hi);
    if (!want0 && !want1) return;
    // the owner of a row = how many strip boundaries lie at or below it: a loop over the (wave-uniform) boundaries, so that
    // the table is read with a scalar index (a per-lane index into the kernel arguments is a waterfall loop)
    uint32_t o0 = 0, o1 = 0;
    for (uint32_t k = 1; k < a.world; k++) { o0 += (r0 << shift) >= a.bounds[k] ? 1u : 0u; o1 += (r1 << shift) >= a.bounds[k] ? 1u : 0u; }
    if (want0 && want1 && o0 == o1) { code[n] = r0 | (x << 14) | VKR_HIT_BOTH_ROWS | tag; owner[n++] = o0; return; }
    if (want0) { code[n] = r0 | (x << 14) | tag; owner[n++] = o0; }
    if (want1) { code[n] = r1 | (x << 14) | tag; owner[n++] = o1; }
  };
  if (have) {
    // texture(albedo, uv): vkr_device.hpp bilinear_taps_u32 — texel rows clamp(y0, y0 + 1), pair (xs, xs + 1)
    const float fx = cfma(uv.x, (float)a.aw, -0.5f), fy = cfma(uv.y, (float)a.ah, -0.5f);
    const int x0 = f2i(floorf(fx)), y0 = f2i(floorf(fy));
    emit((uint32_t)iclamp(y0, 0, a.ah - 1), (uint32_t)iclamp(y0 + 1, 0, a.ah - 1), (uint32_t)iclamp(x0, 0, a.aw - 2), a.win0, a.win1, 0u, 0u);
  }
  if (r.pending != 0u) {
    // texture(normal, hit uv) of the deferred test: sample<FmtRG16U>() on the half-res frame, the uv as the trace had it
    const float2 hv = r.hit_uv;
    const float fx = cfma(hv.x, (float)a.nw, -0.5f), fy = cfma(hv.y, (float)a.nh, -0.5f);
    const int x0 = f2i(floorf(fx)), y0 = f2i(floorf(fy));
    emit((uint32_t)iclamp(y0, 0, a.nh - 1), (uint32_t)iclamp(y0 + 1, 0, a.nh - 1), (uint32_t)iclamp(x0, 0, a.nw - 2), a.nrow0, a.nrow1, VKR_HIT_NORMAL, 1u);
  }
  return n;
}

VKR_DEV int wave_rank_of(uint64_t mask) {  // set bits of `mask` below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__global__ __launch_bounds__(1024) void k_hit_requests(HitReqArgs a, HitTile g) {
  __shared__ uint32_t s_n[HIT_MAX_WORLD], s_base[HIT_MAX_WORLD], s_end[HIT_MAX_WORLD];  // s_end: one past the last slot of the owner's segment
  __shared__ uint32_t s_wcnt[16][HIT_MAX_WORLD];  // pass 2: requests per wave and owner of the tile in hand
  const int tid = threadIdx.x;
  const int tiles = g.tiles_x * g.tiles_y;
  if (tid < HIT_MAX_WORLD) { s_n[tid] = 0u; s_base[tid] = a.seg[tid]; s_end[tid] = a.cap[tid] == 0xFFFFFFFFu ? 0xFFFFFFFFu : a.seg[tid] + a.cap[tid]; }
  __syncthreads();
  if (a.out) {  // pass 2: my base per owner = the segment's start + everything the blocks before me counted in pass 1
    const uint32_t o = (uint32_t)tid & (HIT_MAX_WORLD - 1);
    uint32_t before = 0u;
    for (uint32_t b = (uint32_t)tid / HIT_MAX_WORLD; b < blockIdx.x; b += 1024u / HIT_MAX_WORLD) before += a.workspace[b * HIT_MAX_WORLD + o];
    if (before) atomicAdd(&s_base[o], before);
    __syncthreads();
  }
  uint32_t code[4], owner[4];
  const int wave = tid >> 6;
  for (int first = blockIdx.x; first < tiles; first += gridDim.x * HIT_BATCH) {
    HitRay r[HIT_BATCH];
#pragma unroll
    for (int j = 0; j < HIT_BATCH; j++) r[j] = hit_load(a, g, first + j * gridDim.x, tid);
    // second batch of loads, all in flight together: the hit uv of the pending rays (a load per ray inside the loop below would
    // put up to HIT_BATCH more memory round trips on every pass over the tiles)
#pragma unroll
    for (int j = 0; j < HIT_BATCH; j++) {
      r[j].hit_uv = make_float2(0.0f, 0.0f);
      if (r[j].pending != 0u) r[j].hit_uv = *(const float2*)(texel_ptr<float4>(a.pend_data, 2 * r[j].lx, r[j].ly) + 1);
    }
#pragma unroll
    for (int j = 0; j < HIT_BATCH; j++) {
      const int n = hit_emit(a, r[j], code, owner);
      if (!a.out) {  // pass 1 counts: any order will do
        for (int k = 0; k < n; k++) atomicAdd(&s_n[owner[k]], 1u);
        continue;
      }
      if (__syncthreads_or(n) == 0) continue;  // (most tiles ask for nothing: one barrier instead of the bookkeeping below)
      // Pass 2 hands out the slots of a segment in a FIXED order — tile, wave, request k of the ray, lane — so that the same
      // rays always produce the same bytes on the wire (a frame can be replayed; two runs can be compared request by request).
      // Per tile: every wave counts its requests per owner (ballots), then takes its range behind the waves before it.
      // (a wave without a request writes zeros and skips the ballots; a wave with requests visits only the owners its lanes name)
      const bool wave_has = __ballot(n > 0) != 0ull;
      if ((tid & 63) < (int)a.world) s_wcnt[wave][tid & 63] = 0u;
      if (wave_has) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint64_t todo = __ballot(k < n);
          while (todo) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)owner[k], (int)__builtin_ctzll(todo));
            const uint64_t m = __ballot(k < n && owner[k] == o);
            if ((tid & 63) == 0) s_wcnt[wave][o] += (uint32_t)__popcll(m);
            todo &= ~m;
          }
        }
      }
      __syncthreads();
      uint32_t total = 0u;  // (thread o: what the whole block adds to owner o with this tile)
      if (tid < (int)a.world)
        for (int w = 0; w < 16; w++) total += s_wcnt[w][tid];
      if (wave_has) {
        // lane o of the wave: where owner o's requests of this wave start (the segment's base, what the block has written so far,
        // the waves before this one); within the wave: request k of the ray, then lane
        uint32_t my_at = 0u;
        if ((tid & 63) < (int)a.world) {
          const int o = tid & 63;
          my_at = s_base[o] + s_n[o];
          for (int w = 0; w < wave; w++) my_at += s_wcnt[w][o];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint64_t todo = __ballot(k < n);
          while (todo) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)owner[k], (int)__builtin_ctzll(todo));
            const bool mine = k < n && owner[k] == o;
            const uint64_t m = __ballot(mine);
            const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)my_at, (int)o);
            if (mine) {
              const uint32_t slot = at + (uint32_t)wave_rank_of(m);
              if (slot < s_end[o]) a.out[slot] = code[k];
              else if (a.dropped) *a.dropped = 1u;  // a bounded segment drops what does not fit, and says so
            }
            if ((tid & 63) == (int)o) my_at += (uint32_t)__popcll(m);
            todo &= ~m;
          }
        }
      }
      __syncthreads();
      if (tid < (int)a.world) s_n[tid] += total;
    }
  }
  if (a.out) return;
  __syncthreads();
  if (tid < HIT_MAX_WORLD) {
    if (a.workspace) a.workspace[blockIdx.x * HIT_MAX_WORLD + tid] = s_n[tid];
    if (tid < (int)a.world && s_n[tid]) atomicAdd(&a.counts[tid], s_n[tid]);
  }
}

// the texel pair of the requested row — and of the row below it for a two-row request — from the owner's window image of
// the surface the request names: 16 bytes back
__global__ __launch_bounds__(256) void k_hit_reply(Tex albedo, Tex normals, uint32_t has_normals, const vkr_hit_request* req, uint32_t count,
                                                   uint4* replies, uint32_t* errors) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const uint32_t r = req[i];
  if (r == VKR_HIT_NO_REQUEST) { replies[i] = make_uint4(0u, 0u, 0u, 0u); return; }  // an unused slot of a fixed-capacity segment
  const bool nrm = (r & VKR_HIT_NORMAL) != 0u, both = (r & VKR_HIT_BOTH_ROWS) != 0u;
  const Tex& t = nrm ? normals : albedo;
  const int ly = (int)(r & 0x3FFFu) - t.oy, lx = (int)((r >> 14) & 0x3FFFu) - t.ox;
  if ((nrm && !has_normals) || ly < 0 || ly + (both ? 1 : 0) >= t.h || lx < 0 || lx + 1 >= t.w) {  // not mine: the ranks' strips disagree
    atomicAdd(errors, 1u);
    errors[1] = r; errors[2] = i;  // (diagnostics: one of the offending requests and its slot)
    replies[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const U32x2 v0 = load_u32x2(t.p + toff(t, lx, ly, 4));
  const U32x2 v1 = both ? load_u32x2(t.p + toff(t, lx, ly + 1, 4)) : v0;
  replies[i] = make_uint4(v0.x, v0.y, v1.x, v1.y);
}

__global__ __launch_bounds__(256) void k_hit_scatter(Tex frame_albedo, Tex frame_normals, const vkr_hit_request* req, const uint4* replies, uint32_t count) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const uint32_t r = req[i];
  if (r == VKR_HIT_NO_REQUEST) return;
  const uint4 v = replies[i];
  const Tex& t = (r & VKR_HIT_NORMAL) ? frame_normals : frame_albedo;
  const int row = (int)(r & 0x3FFFu), x = (int)((r >> 14) & 0x3FFFu);
  uint32_t* dst = (uint32_t*)(const_cast<uint8_t*>(t.p) + toff(t, x, row, 4));
  dst[0] = v.x; dst[1] = v.y;
  if (r & VKR_HIT_BOTH_ROWS) {
    uint32_t* dst1 = (uint32_t*)(const_cast<uint8_t*>(t.p) + toff(t, x, row + 1, 4));
    dst1[0] = v.z; dst1[1] = v.w;
  }
}

}  // namespace vkr

using namespace vkr;

static int hit_requests(const vkr_hit_sources* src, const uint32_t* row_bounds, uint32_t world, uint32_t* counts, uint32_t* workspace,
                        const uint32_t* segments, const uint32_t* capacities, vkr_hit_request* out, uint32_t* dropped, void* stream) {
  if (!src || !row_bounds || (!out && !counts) || (out && (!workspace || !segments))) { set_error("hit_requests: NULL argument"); return VKR_ERR_NULL; }
  if (world < 1 || world > HIT_MAX_WORLD) { set_error("hit_requests: world %u (1..%d)", world, HIT_MAX_WORLD); return VKR_ERR_EXTENT; }
  if (src->albedo_width > 16384 || src->albedo_height > 16384) { set_error("hit_requests: a request packs row and column into 14 bits each (frame <= 16384)"); return VKR_ERR_EXTENT; }
  if (src->albedo_width < 2 || src->albedo_height < 1 || row_bounds[0] != 0 || row_bounds[world] != src->albedo_height ||
      src->window_row0 >= src->window_row1 || src->window_row1 > src->albedo_height) {
    set_error("hit_requests: frame %ux%u, window rows [%u, %u), bounds [%u .. %u]", src->albedo_width, src->albedo_height, src->window_row0,
              src->window_row1, row_bounds[0], row_bounds[world]);
    return VKR_ERR_EXTENT;
  }
  HitReqArgs a;
  VKR_TRY(make_tex(src->rays, 0, VKR_FMT_RGBA16_UNORM, "hit_requests.rays", &a.rays));
  a.aw = (int)src->albedo_width; a.ah = (int)src->albedo_height;
  for (uint32_t r = 0; r <= world; r++) {
    if ((r && row_bounds[r] <= row_bounds[r - 1]) || (row_bounds[r] & 1u)) { set_error("hit_requests: row bounds must increase and be even"); return VKR_ERR_EXTENT; }
    a.bounds[r] = row_bounds[r];
  }
  for (uint32_t r = world + 1; r <= HIT_MAX_WORLD; r++) a.bounds[r] = src->albedo_height;
  a.world = world; a.win0 = src->window_row0; a.win1 = src->window_row1;
  a.has_normals = src->pending_mask ? 1u : 0u;
  a.pend_mask = a.rays; a.pend_data = a.rays; a.nw = 2; a.nh = 1; a.nrow0 = 0; a.nrow1 = 1;
  if (a.has_normals) {
    VKR_TRY(make_tex(src->pending_mask, 0, VKR_FMT_R8_UNORM, "hit_requests.pending_mask", &a.pend_mask));
    VKR_TRY(make_tex(src->pending_data, 0, VKR_FMT_RGBA32_SFLOAT, "hit_requests.pending_data", &a.pend_data));
    if (a.pend_mask.w != a.rays.w || a.pend_mask.h != a.rays.h || a.pend_data.w != 2 * a.rays.w || a.pend_data.h != a.rays.h ||
        src->normal_width < 2 || src->normal_row0 >= src->normal_row1 || src->normal_row1 > src->normal_height ||
        2 * src->normal_height > src->albedo_height) {
      set_error("hit_requests: pending images / normal rows do not match the rays");
      return VKR_ERR_EXTENT;
    }
    a.nw = (int)src->normal_width; a.nh = (int)src->normal_height; a.nrow0 = src->normal_row0; a.nrow1 = src->normal_row1;
  }
  a.counts = counts; a.workspace = workspace; a.out = out; a.dropped = dropped;
  for (uint32_t r = 0; r < HIT_MAX_WORLD; r++) {
    a.seg[r] = (out && r < world) ? segments[r] : 0u;
    a.cap[r] = (out && capacities && r < world) ? capacities[r] : 0xFFFFFFFFu;
  }
  HitTile g;
  g.tiles_x = (a.rays.w + 63) / 64;
  g.tiles_y = (a.rays.h + HIT_TILE_H - 1) / HIT_TILE_H + 1;  // one more row of tiles: see hit_load
  const int tiles = g.tiles_x * g.tiles_y;
  hipLaunchKernelGGL(k_hit_requests, dim3((uint32_t)(tiles < HIT_BLOCKS ? tiles : HIT_BLOCKS)), dim3(1024), 0, (hipStream_t)stream, a, g);
  return launch_status("hit_requests");
}

extern "C" int vkr_hit_requests(const vkr_hit_sources* src, const uint32_t* row_bounds, uint32_t world, uint32_t* counts, uint32_t* workspace,
                                const uint32_t* segments, vkr_hit_request* out, void* stream) {
  return hit_requests(src, row_bounds, world, counts, workspace, segments, nullptr, out, nullptr, stream);
}

extern "C" int vkr_hit_requests_bounded(const vkr_hit_sources* src, const uint32_t* row_bounds, uint32_t world, uint32_t* workspace,
                                        const uint32_t* segments, const uint32_t* capacities, vkr_hit_request* out, uint32_t* dropped_flag,
                                        void* stream) {
  if (!out || !capacities || !dropped_flag) { set_error("hit_requests_bounded: NULL argument"); return VKR_ERR_NULL; }
  return hit_requests(src, row_bounds, world, nullptr, workspace, segments, capacities, out, dropped_flag, stream);
}

extern "C" int vkr_hit_reply(const vkr_img* albedo, const vkr_img* normals, const vkr_hit_request* requests, uint32_t count, void* replies,
                             uint32_t* error_counter, void* stream) {
  if (count == 0) return VKR_OK;
  if (!requests || !replies || !error_counter) { set_error("hit_reply: NULL argument"); return VKR_ERR_NULL; }
  Tex t, n;
  VKR_TRY(make_tex(albedo, 0, VKR_FMT_RGBA8_SRGB, "hit_reply.albedo", &t));
  n = t;
  if (normals) VKR_TRY(make_tex(normals, 0, VKR_FMT_RG16_UNORM, "hit_reply.normals", &n));
  hipLaunchKernelGGL(k_hit_reply, dim3((count + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, t, n, normals ? 1u : 0u, requests, count, (uint4*)replies, error_counter);
  return launch_status("hit_reply");
}

extern "C" int vkr_hit_scatter(const vkr_img* frame_albedo, const vkr_img* frame_normals, const vkr_hit_request* requests, const void* replies,
                               uint32_t count, void* stream) {
  if (count == 0) return VKR_OK;
  if (!requests || !replies) { set_error("hit_scatter: NULL argument"); return VKR_ERR_NULL; }
  Tex t, n;
  VKR_TRY(make_tex(frame_albedo, 0, VKR_FMT_RGBA8_SRGB, "hit_scatter.frame_albedo", &t));
  if (t.ox != 0 || t.oy != 0 || t.w != t.fw || t.h != t.fh) { set_error("hit_scatter: the destination must be the whole-frame image"); return VKR_ERR_EXTENT; }
  n = t;
  if (frame_normals) {
    VKR_TRY(make_tex(frame_normals, 0, VKR_FMT_RG16_UNORM, "hit_scatter.frame_normals", &n));
    if (n.ox != 0 || n.oy != 0 || n.w != n.fw || n.h != n.fh) { set_error("hit_scatter: the destination must be the whole-frame image"); return VKR_ERR_EXTENT; }
  }
  hipLaunchKernelGGL(k_hit_scatter, dim3((count + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, t, n, requests, (const uint4*)replies, count);
  return launch_status("hit_scatter");
}
