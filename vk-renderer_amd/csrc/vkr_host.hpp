// vkr_host.hpp — host side of the C-ABI shim: descriptor validation, vkr_img -> device
// view conversion, launch error capture.  Shared by the per-pass .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include "../../include/vkr_postfx.h"
#include "vkr_device.hpp"

namespace vkr {

void set_error(const char* fmt, ...);
uint32_t switches();  // VKR_SWITCH_* (common.hip): the environment read once, then vkr_set_switches()

enum {
  VKR_OK = 0,
  VKR_ERR_NULL = 1001,      // a required descriptor / pointer is NULL
  VKR_ERR_FORMAT = 1002,    // an image has a format the program cannot bind
  VKR_ERR_EXTENT = 1003,    // extents / windows of the bound images do not agree
  VKR_ERR_MIPS = 1004,      // not enough mips (downsample_pass.cpp:37-39)
  VKR_ERR_LAYOUT = 1005     // pitch / offset / alignment of a descriptor is unusable
};

inline int mip_dim(uint32_t v, int i) { int r = (int)(v >> i); return r > 0 ? r : 1; }

// validated conversion of one mip of a descriptor
inline int make_tex(const vkr_img* d, int mip, uint32_t want_format, const char* what, Tex* out) {
  if (!d || !d->base) { set_error("%s: NULL image", what); return VKR_ERR_NULL; }
  if (d->format != want_format) { set_error("%s: format %u, expected %u", what, d->format, want_format); return VKR_ERR_FORMAT; }
  if (mip < 0 || mip >= (int)d->mip_count || d->mip_count > VKR_MAX_MIPS) { set_error("%s: mip %d not in view (%u mips)", what, mip, d->mip_count); return VKR_ERR_MIPS; }
  uint32_t bpp = vkr_format_bytes(d->format);
  int w = mip_dim(d->width, mip), h = mip_dim(d->height, mip);
  if (d->width == 0 || d->height == 0 || d->pitch_bytes[mip] < (uint32_t)w * bpp || (d->pitch_bytes[mip] % bpp) != 0 ||
      (((uintptr_t)d->base + d->mip_offset[mip]) % (bpp < 4 ? bpp : 4)) != 0) {
    set_error("%s: bad layout (w %d pitch %u)", what, w, d->pitch_bytes[mip]);
    return VKR_ERR_LAYOUT;
  }
  // texel offsets are 32-bit, rows x pitch a 24-bit multiply (toff() in vkr_device.hpp)
  if (d->pitch_bytes[mip] >= (1u << 24) || h >= (1 << 24) || (uint64_t)d->pitch_bytes[mip] * (uint64_t)h >= (1ull << 32)) {
    set_error("%s: window too large for 32-bit texel offsets (pitch %u x %d rows)", what, d->pitch_bytes[mip], h);
    return VKR_ERR_LAYOUT;
  }
  int fw = mip_dim(d->full_width, mip), fh = mip_dim(d->full_height, mip);
  int ox = d->origin_x >> mip, oy = d->origin_y >> mip;
  if (mip > 0 && d->origin_x >= 0 && d->origin_y >= 0) {
    // Mip extents are floor(extent / 2^mip), at least 1: where the frame's extent is not a multiple of 2^mip, a window
    // that starts in the last, partial texel of a coarse mip (or is shorter than one texel of it) would start past the
    // mip's last row.  Such a level holds no whole texel of the window; it is placed on the frame's last row / column so
    // that the chain can still be built (nothing reads it: a window's levels past the gathered ones are never sampled).
    if (ox + w > fw && w <= fw) ox = fw - w;
    if (oy + h > fh && h <= fh) oy = fh - h;
  }
  if (d->origin_x < 0 || d->origin_y < 0 || ox + w > fw || oy + h > fh) {
    set_error("%s: window (%d,%d)+(%d,%d) outside frame %dx%d at mip %d", what, ox, oy, w, h, fw, fh, mip);
    return VKR_ERR_EXTENT;
  }
  out->p = (const uint8_t*)d->base + d->mip_offset[mip];
  out->pitch = (int)d->pitch_bytes[mip];
  out->w = w; out->h = h; out->fw = fw; out->fh = fh; out->ox = ox; out->oy = oy;
  return VKR_OK;
}

inline bool same_window(const Tex& a, const Tex& b) {
  return a.w == b.w && a.h == b.h && a.fw == b.fw && a.fh == b.fh && a.ox == b.ox && a.oy == b.oy;
}

// same window AND row pitch: texel (x, y) has the same byte offset in both (4-byte formats share bilinear footprints)
inline bool same_layout(const Tex& a, const Tex& b) { return same_window(a, b) && a.pitch == b.pitch; }

inline void load_mat(Mat4& dst, const vkr_mat4& src) { std::memcpy(dst.m, src.m, sizeof(float) * 16); }

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("%s: launch failed: %s", what, hipGetErrorString(e)); return (int)e; }
  return VKR_OK;
}

#define VKR_TRY(expr) do { int _rc = (expr); if (_rc != VKR_OK) return _rc; } while (0)

inline dim3 grid2d(int w, int h, dim3 block) { return dim3((w + block.x - 1) / block.x, (h + block.y - 1) / block.y, 1); }

}  // namespace vkr
