// exchange.hip — "copy_rects": the pack / scatter step of the multi-GPU exchanges (SURVEY.md 8(e);
// no reference counterpart, the reference is single-GPU).  One launch moves up to VKR_MAX_RECTS
// pitch-linear byte rectangles: a tile's surfaces into the send buffer of an all-gather, the
// gathered tiles of every rank into the whole-frame images, the halo rings of a history surface
// into / out of the per-neighbour buffers.  HBM-bound byte moving: 16-byte words when every
// rectangle of the batch allows it, 4-byte words otherwise; blockIdx.y = rectangle, blockIdx.x
// strides over its words.
#include "vkr_host.hpp"

namespace vkr {

struct RectBatch { vkr_rect_copy r[VKR_MAX_RECTS]; };

template <typename W>
__global__ __launch_bounds__(256) void k_copy_rects(RectBatch b) {
  const vkr_rect_copy rc = b.r[blockIdx.y];
  const uint32_t words_per_row = rc.row_bytes / (uint32_t)sizeof(W);
  const uint32_t total = words_per_row * rc.rows;
  const uint8_t* src = (const uint8_t*)rc.src;
  uint8_t* dst = (uint8_t*)rc.dst;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t row = i / words_per_row, col = i - row * words_per_row;
    const W v = *(const W*)(src + (uint64_t)row * rc.src_pitch + (uint64_t)col * sizeof(W));
    *(W*)(dst + (uint64_t)row * rc.dst_pitch + (uint64_t)col * sizeof(W)) = v;
  }
}

}  // namespace vkr

using namespace vkr;

extern "C" int vkr_copy_rects(const vkr_rect_copy* rects, uint32_t count, void* stream) {
  if (count == 0) return VKR_OK;
  if (!rects) { set_error("copy_rects: NULL rectangle list"); return VKR_ERR_NULL; }
  for (uint32_t first = 0; first < count; first += VKR_MAX_RECTS) {
    const uint32_t n = count - first < VKR_MAX_RECTS ? count - first : VKR_MAX_RECTS;
    RectBatch b;
    uint64_t align = 0, most = 0;
    for (uint32_t i = 0; i < n; i++) {
      const vkr_rect_copy& r = rects[first + i];
      if (!r.src || !r.dst) { set_error("copy_rects: rectangle %u has a NULL address", first + i); return VKR_ERR_NULL; }
      if (r.row_bytes == 0 || r.rows == 0 || r.src_pitch < r.row_bytes || r.dst_pitch < r.row_bytes ||
          (uint64_t)r.row_bytes * r.rows > 0xFFFFFFFFull) {
        set_error("copy_rects: rectangle %u: %u rows of %u bytes, pitches %u / %u", first + i, r.rows, r.row_bytes, r.src_pitch, r.dst_pitch);
        return VKR_ERR_EXTENT;
      }
      align |= r.src | r.dst | r.src_pitch | r.dst_pitch | r.row_bytes;
      const uint64_t bytes = (uint64_t)r.row_bytes * r.rows;
      most = bytes > most ? bytes : most;
      b.r[i] = r;
    }
    if (align % 4 != 0) { set_error("copy_rects: addresses, pitches and row lengths must be multiples of 4 bytes"); return VKR_ERR_LAYOUT; }
    const uint32_t word = align % 16 == 0 ? 16 : 4;
    // enough blocks for the largest rectangle to give every thread ~4 words, at most 1024 per rectangle
    uint64_t blocks = (most / word + 256 * 4 - 1) / (256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    const dim3 grid((uint32_t)blocks, n);
    if (word == 16) hipLaunchKernelGGL(k_copy_rects<uint4>, grid, dim3(256), 0, (hipStream_t)stream, b);
    else hipLaunchKernelGGL(k_copy_rects<uint32_t>, grid, dim3(256), 0, (hipStream_t)stream, b);
    VKR_TRY(launch_status("copy_rects"));
  }
  return VKR_OK;
}
