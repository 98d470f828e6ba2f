"""Pitch-linear image buffers in the reference's storage formats.

An ImageBuf owns one allocation holding every mip of an image (rows 256-B aligned) and
hands out `VkrImg` view descriptors for the C-ABI.  The backing store is either a numpy
array (host memory: fixtures, checkers) or a torch uint8 tensor (device memory: torch is only the
allocator here).  `decode()` turns storage into float arrays for comparisons in tests.
"""
import ctypes as C

import numpy as np

from . import abi

ROW_ALIGN = 256


def _align(v, a):
    return (v + a - 1) // a * a


def mip_extent(v, i):
    return max(1, v >> i)


class ImageBuf:
    def __init__(self, fmt, width, height, mips=1, device=None, full=None, origin=(0, 0), fill=0):
        """width/height: extent of the window held in memory (mip 0); full: frame extent."""
        self.format = fmt
        self.width, self.height, self.mips = int(width), int(height), int(mips)
        self.full = (int(full[0]), int(full[1])) if full else (self.width, self.height)
        self.origin = (int(origin[0]), int(origin[1]))
        self.bpp = abi.FORMAT_BYTES[fmt]
        self.pitch, self.offset = [], []
        off = 0
        for i in range(self.mips):
            p = _align(mip_extent(self.width, i) * self.bpp, ROW_ALIGN)
            self.pitch.append(p)
            self.offset.append(off)
            off += _align(p * mip_extent(self.height, i), ROW_ALIGN)
        self.nbytes = off
        self.device = device
        if device is None:
            self.host = np.full(self.nbytes, fill, dtype=np.uint8)
            self.tensor = None
            self.ptr = self.host.ctypes.data
        else:
            import torch

            self.tensor = torch.full((self.nbytes,), fill, dtype=torch.uint8, device=device)
            self.host = None
            self.ptr = self.tensor.data_ptr()

    # ---- descriptors ------------------------------------------------------------
    def desc(self, base_mip=0, mip_count=None):
        """View descriptor starting at `base_mip` (what builder.sample_image(.., base_mip, count) binds)."""
        if mip_count is None:
            mip_count = self.mips - base_mip
        assert 0 <= base_mip and base_mip + mip_count <= self.mips
        d = abi.VkrImg()
        d.base = self.ptr + self.offset[base_mip]
        d.format = self.format
        d.mip_count = mip_count
        d.width, d.height = mip_extent(self.width, base_mip), mip_extent(self.height, base_mip)
        d.full_width, d.full_height = mip_extent(self.full[0], base_mip), mip_extent(self.full[1], base_mip)
        d.origin_x, d.origin_y = self.origin[0] >> base_mip, self.origin[1] >> base_mip
        for i in range(mip_count):
            d.pitch_bytes[i] = self.pitch[base_mip + i]
            d.mip_offset[i] = self.offset[base_mip + i] - self.offset[base_mip]
        return d

    # ---- host access --------------------------------------------------------------
    def to_host(self):
        if self.host is not None:
            return self.host
        return self.tensor.cpu().numpy()

    def upload(self, host_bytes):
        host_bytes = np.ascontiguousarray(host_bytes, dtype=np.uint8).reshape(-1)
        assert host_bytes.size == self.nbytes
        if self.host is not None:
            self.host[:] = host_bytes
        else:
            import torch

            self.tensor.copy_(torch.from_numpy(host_bytes))

    def copy_from(self, other):
        assert other.nbytes == self.nbytes
        self.upload(other.to_host())

    def raw(self, mip=0, host=None):
        """Raw storage of one mip as an array [h, w, channels] of the storage dtype."""
        host = self.to_host() if host is None else host
        w, h = mip_extent(self.width, mip), mip_extent(self.height, mip)
        rows = host[self.offset[mip]: self.offset[mip] + self.pitch[mip] * h].reshape(h, self.pitch[mip])[:, : w * self.bpp]
        f = self.format
        if f == abi.FMT_D24_UNORM_S8:
            return np.ascontiguousarray(rows).view(np.uint32).reshape(h, w, 1)
        if f in (abi.FMT_RG16_UNORM,):
            return np.ascontiguousarray(rows).view(np.uint16).reshape(h, w, 2)
        if f == abi.FMT_RG16_SFLOAT:
            return np.ascontiguousarray(rows).view(np.float16).reshape(h, w, 2)
        if f in (abi.FMT_RGBA8_SRGB, abi.FMT_RGBA8_UNORM):
            return np.ascontiguousarray(rows).reshape(h, w, 4)
        if f == abi.FMT_RGBA16_UNORM:
            return np.ascontiguousarray(rows).view(np.uint16).reshape(h, w, 4)
        if f == abi.FMT_RGBA16_SFLOAT:
            return np.ascontiguousarray(rows).view(np.float16).reshape(h, w, 4)
        if f == abi.FMT_R16_SFLOAT:
            return np.ascontiguousarray(rows).view(np.float16).reshape(h, w, 1)
        if f == abi.FMT_R32_SFLOAT:
            return np.ascontiguousarray(rows).view(np.float32).reshape(h, w, 1)
        if f == abi.FMT_R8_UNORM:
            return np.ascontiguousarray(rows).reshape(h, w, 1)
        raise ValueError(f)

    def set_raw(self, arr, mip=0):
        """Write raw storage values (array [h, w, c] of the storage dtype) into one mip (host-backed only)."""
        assert self.host is not None
        w, h = mip_extent(self.width, mip), mip_extent(self.height, mip)
        a = np.ascontiguousarray(arr).reshape(h, -1).view(np.uint8)
        assert a.shape[1] == w * self.bpp, (a.shape, w, self.bpp)
        view = self.host[self.offset[mip]: self.offset[mip] + self.pitch[mip] * h].reshape(h, self.pitch[mip])
        view[:, : w * self.bpp] = a

    def decode(self, mip=0, host=None):
        """float32 [h, w, c] of what a texelFetch returns (sRGB decoded as plain UNORM8 codes /255 is NOT applied:
        sRGB images decode through the sRGB EOTF)."""
        r = self.raw(mip, host)
        f = self.format
        if f == abi.FMT_D24_UNORM_S8:
            return ((r & 0xFFFFFF).astype(np.float32) / np.float32(16777215.0)).astype(np.float32)
        if f in (abi.FMT_RG16_UNORM, abi.FMT_RGBA16_UNORM):
            return (r.astype(np.float32) / np.float32(65535.0)).astype(np.float32)
        if f in (abi.FMT_RGBA8_UNORM, abi.FMT_R8_UNORM):
            return (r.astype(np.float32) / np.float32(255.0)).astype(np.float32)
        if f == abi.FMT_RGBA8_SRGB:
            c = r.astype(np.float64) / 255.0
            lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
            lin[..., 3] = c[..., 3]
            return lin.astype(np.float32)
        return r.astype(np.float32)


def depth_mip_count(width, height):
    """scene_renderer.cpp:13: floor(log2(max(w, h))) + 1"""
    return int(np.floor(np.log2(max(width, height)))) + 1


def ptr(obj):
    return C.byref(obj)
