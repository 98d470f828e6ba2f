"""Error behaviour of the C-ABI (include/vkr_postfx.h): every entry validates its descriptors on the host before it
launches anything, so a bad binding is refused with a code and a message — the counterpart of the reference's
`throw std::runtime_error` in the pass constructors / `DescriptorSet` checks (e.g. downsample_pass.cpp:37-39).
No GPU needed: each call below fails validation and never reaches a launch."""
import ctypes as C

import pytest

from vk_renderer_amd import abi

ERR_NULL, ERR_FORMAT, ERR_EXTENT, ERR_MIPS, ERR_LAYOUT = 1001, 1002, 1003, 1004, 1005
FAKE = 0x10000  # a non-NULL base: validation never dereferences it


def img(fmt, w, h, mips=1, base=FAKE, pitch=None, full=None, origin=(0, 0)):
    d = abi.VkrImg()
    d.base = base
    d.format = fmt
    d.mip_count = mips
    d.width, d.height = w, h
    d.full_width, d.full_height = full or (w, h)
    d.origin_x, d.origin_y = origin
    bpp = abi.product().vkr_format_bytes(fmt)
    off = 0
    for m in range(mips):
        mw, mh = max(w >> m, 1), max(h >> m, 1)
        d.pitch_bytes[m] = pitch if (pitch is not None and m == 0) else mw * bpp
        d.mip_offset[m] = off
        off += d.pitch_bytes[m] * mh
    return d


@pytest.fixture(scope="module")
def lib():
    return abi.product()


def taa(lib, **over):
    w, h = 64, 32
    d = dict(history=img(abi.FMT_RGBA16_SFLOAT, w, h), hist_depth=img(abi.FMT_D24_UNORM_S8, w, h), depth=img(abi.FMT_D24_UNORM_S8, w, h),
             velocity=img(abi.FMT_RG16_SFLOAT, w, h), color=img(abi.FMT_RGBA8_SRGB, w, h), out=img(abi.FMT_RGBA16_SFLOAT, w, h),
             params=abi.ReprojectParams())
    d.update(over)
    ref = lambda v: None if v is None else C.byref(v)  # noqa: E731
    rc = lib.vkr_taa_resolve(ref(d["history"]), ref(d["hist_depth"]), ref(d["depth"]), ref(d["velocity"]), ref(d["color"]), ref(d["out"]),
                             ref(d["params"]), None)
    return rc, (lib.vkr_last_error() or b"").decode()


def test_null_descriptor_and_null_params(lib):
    rc, msg = taa(lib, history=None)
    assert rc == ERR_NULL and "taa_resolve.history" in msg
    rc, msg = taa(lib, history=img(abi.FMT_RGBA16_SFLOAT, 64, 32, base=None))
    assert rc == ERR_NULL and "NULL image" in msg
    rc, msg = taa(lib, params=None)
    assert rc == ERR_NULL and "params" in msg


def test_wrong_format_names_the_binding(lib):
    rc, msg = taa(lib, velocity=img(abi.FMT_RG16_UNORM, 64, 32))
    assert rc == ERR_FORMAT and "taa_resolve.velocity" in msg and "format" in msg


def test_layout_rules(lib):
    rc, msg = taa(lib, color=img(abi.FMT_RGBA8_SRGB, 64, 32, pitch=64 * 4 - 4))  # pitch shorter than a row
    assert rc == ERR_LAYOUT and "bad layout" in msg
    rc, msg = taa(lib, color=img(abi.FMT_RGBA8_SRGB, 64, 32, pitch=64 * 4 + 2))  # not a multiple of the texel size
    assert rc == ERR_LAYOUT
    rc, msg = taa(lib, color=img(abi.FMT_RGBA8_SRGB, 64, 32, base=FAKE + 2))      # misaligned base
    assert rc == ERR_LAYOUT
    # 32-bit texel offsets: pitch < 16 MiB and pitch x rows < 4 GiB (vkr_img.pitch_bytes in the header)
    rc, msg = taa(lib, color=img(abi.FMT_RGBA8_SRGB, 64, 32, pitch=1 << 24))
    assert rc == ERR_LAYOUT and "32-bit texel offsets" in msg
    rc, msg = taa(lib, color=img(abi.FMT_RGBA8_SRGB, 64, 1 << 20, pitch=1 << 13))   # 8 KiB x 1 Mi rows = 8 GiB
    assert rc == ERR_LAYOUT and "32-bit texel offsets" in msg


def test_window_must_lie_inside_the_frame(lib):
    rc, msg = taa(lib, depth=img(abi.FMT_D24_UNORM_S8, 64, 32, full=(64, 40), origin=(0, 16)))
    assert rc == ERR_EXTENT and "outside frame" in msg
    rc, msg = taa(lib, depth=img(abi.FMT_D24_UNORM_S8, 64, 32, full=(64, 64), origin=(0, -2)))
    assert rc == ERR_EXTENT


def test_mip_rules(lib):
    bad = img(abi.FMT_D24_UNORM_S8, 64, 32, mips=1)
    bad.mip_count = abi.VKR_MAX_MIPS + 1
    assert lib.vkr_depth_mips(C.byref(bad), 0, None) == ERR_MIPS
    assert lib.vkr_depth_mips(None, 0, None) == ERR_NULL
    # a one-mip view has nothing to build: a no-op, not an error (the >= 2 mips rule of downsample_pass.cpp:37-39 is
    # enforced where the reference enforces it, in the DownsamplePass constructor of the host mirror)
    assert lib.vkr_depth_mips(C.byref(img(abi.FMT_D24_UNORM_S8, 64, 32, mips=1)), 0, None) == 0
    # the G-buffer downsample reads view mip 1 of the depth image: a one-mip view is refused
    d1 = img(abi.FMT_D24_UNORM_S8, 64, 32, mips=1)
    n, v = img(abi.FMT_RG16_UNORM, 64, 32), img(abi.FMT_RG16_SFLOAT, 64, 32)
    on, ov = img(abi.FMT_RG16_UNORM, 32, 16), img(abi.FMT_RG16_SFLOAT, 32, 16)
    rc = lib.vkr_downsample_gbuffer(C.byref(d1), C.byref(n), C.byref(v), C.byref(on), C.byref(ov), None)
    assert rc == ERR_MIPS and b"mip" in lib.vkr_last_error()


def test_error_string_is_replaced_by_the_next_failure(lib):
    taa(lib, history=None)
    first = lib.vkr_last_error()
    taa(lib, velocity=img(abi.FMT_RG16_UNORM, 64, 32))
    assert lib.vkr_last_error() != first


def test_exchange_entries_refuse_missing_arguments(lib):
    """vkr_all_gather / vkr_all_gather_v / vkr_halo_exchange: nothing to do is fine, a missing communicator is an error
    (before RCCL is even loaded)."""
    for name in ("vkr_all_gather", "vkr_all_gather_v", "vkr_halo_exchange"):
        fn = getattr(lib, name)
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        fn.restype = C.c_int
        assert fn(None, None, 0, None) == 0
        assert fn(None, None, 1, None) == ERR_NULL and b"NULL" in lib.vkr_last_error()
