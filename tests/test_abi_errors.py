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


# ---- the multi-GPU entries of round 3: hit colours / hit normals by request / reply, the windowed trace -----------------
class HitSources(C.Structure):
    _fields_ = [("rays", C.POINTER(abi.VkrImg)), ("albedo_width", C.c_uint32), ("albedo_height", C.c_uint32), ("window_row0", C.c_uint32),
                ("window_row1", C.c_uint32), ("pending_mask", C.POINTER(abi.VkrImg)), ("pending_data", C.POINTER(abi.VkrImg)),
                ("normal_width", C.c_uint32), ("normal_height", C.c_uint32), ("normal_row0", C.c_uint32), ("normal_row1", C.c_uint32)]


def _hit_requests(lib, src, bounds, world, counts=FAKE, cursors=None, segments=None, out=None):
    lib.vkr_hit_requests.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
    b = (C.c_uint32 * len(bounds))(*bounds)
    seg = (C.c_uint32 * world)() if segments else None
    rc = lib.vkr_hit_requests(C.byref(src) if src is not None else None, b, world, counts, cursors, seg, out, None)
    return rc, (lib.vkr_last_error() or b"").decode()


def test_hit_requests_refuse_inconsistent_strips(lib):
    rays = img(abi.FMT_RGBA16_UNORM, 128, 64, full=(128, 256), origin=(0, 32))
    src = HitSources(C.pointer(rays), 256, 512, 64, 192, None, None, 0, 0, 0, 0)
    rc, msg = _hit_requests(lib, None, [0, 256, 512], 2)
    assert rc == ERR_NULL
    rc, msg = _hit_requests(lib, src, [0, 256, 512], 17)
    assert rc == ERR_EXTENT and "world 17" in msg
    rc, msg = _hit_requests(lib, src, [0, 256, 500], 2)          # bounds must end at the frame height
    assert rc == ERR_EXTENT and "bounds" in msg
    rc, msg = _hit_requests(lib, src, [0, 255, 512], 2)          # strips are cut at even rows
    assert rc == ERR_EXTENT and "even" in msg
    rc, msg = _hit_requests(lib, src, [0, 256, 512], 2, counts=None)  # pass 1 needs the counters
    assert rc == ERR_NULL
    rc, msg = _hit_requests(lib, src, [0, 256, 512], 2, out=FAKE)     # pass 2 needs the workspace of pass 1 and the segments
    assert rc == ERR_NULL
    big = HitSources(C.pointer(rays), 32768, 512, 64, 192, None, None, 0, 0, 0, 0)
    rc, msg = _hit_requests(lib, big, [0, 256, 512], 2)
    assert rc == ERR_EXTENT and "14 bits" in msg
    # pending images that do not match the rays
    mask = img(abi.FMT_R8_UNORM, 64, 64)
    data = img(abi.FMT_RGBA32_SFLOAT, 256, 64)
    bad = HitSources(C.pointer(rays), 256, 512, 64, 192, C.pointer(mask), C.pointer(data), 128, 256, 32, 96)
    rc, msg = _hit_requests(lib, bad, [0, 256, 512], 2)
    assert rc == ERR_EXTENT and "pending" in msg


def test_hit_reply_and_scatter_validate_their_images(lib):
    lib.vkr_hit_reply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vkr_hit_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    albedo = img(abi.FMT_RGBA8_SRGB, 256, 128, full=(256, 512), origin=(0, 64))
    assert lib.vkr_hit_reply(C.byref(albedo), None, None, 0, None, None, None) == 0            # nothing to answer
    assert lib.vkr_hit_reply(C.byref(albedo), None, None, 4, FAKE, FAKE, None) == ERR_NULL     # requests missing
    wrong = img(abi.FMT_RGBA8_UNORM, 256, 128)
    assert lib.vkr_hit_reply(C.byref(wrong), None, FAKE, 4, FAKE, FAKE, None) == ERR_FORMAT
    # the scatter target is the WHOLE-frame image
    assert lib.vkr_hit_scatter(C.byref(albedo), None, FAKE, FAKE, 4, None) == ERR_EXTENT
    assert "whole-frame" in (lib.vkr_last_error() or b"").decode()


def test_windowed_trace_and_validate_refuse_bad_windows(lib):
    class WindowPush(C.Structure):
        _fields_ = [("max_roughness", C.c_float), ("normal_row0", C.c_uint32), ("normal_row1", C.c_uint32)]

    lib.vkr_sssr_trace_windowed.argtypes = [C.c_void_p] * 12
    lib.vkr_sssr_validate.argtypes = [C.c_void_p] * 6
    w2, h2, fh = 128, 64, 256
    depth = img(abi.FMT_D24_UNORM_S8, 128, fh, mips=3)
    normal = img(abi.FMT_RG16_UNORM, 128, fh)
    material = img(abi.FMT_RGBA8_SRGB, 256, 128, full=(256, 512), origin=(0, 64))
    rays = img(abi.FMT_RGBA16_UNORM, w2, h2, full=(128, fh), origin=(0, 32))
    occ = img(abi.FMT_RGBA16_SFLOAT, w2, h2, full=(128, fh), origin=(0, 32))
    pdf = img(abi.FMT_R32_SFLOAT, 1024, 1024)
    mask = img(abi.FMT_R8_UNORM, w2, h2, full=(128, fh), origin=(0, 32))
    data = img(abi.FMT_RGBA32_SFLOAT, 2 * w2, h2)
    params = abi.TraceParams()
    halton = (C.c_float * (4 * 128 + 4))()
    hp = (C.addressof(halton) + 15) & ~15

    def call(push, n=normal, m=mask, d=data):
        return lib.vkr_sssr_trace_windowed(C.byref(depth), C.byref(n), C.byref(material), C.byref(params), hp, C.byref(rays), C.byref(occ),
                                           C.byref(pdf), C.byref(m), C.byref(d), C.byref(push) if push is not None else None, None)

    assert call(None) == ERR_NULL
    assert call(WindowPush(1.0, 96, 32)) == ERR_EXTENT                      # rows must be a range inside the frame
    assert call(WindowPush(1.0, 32, fh + 1)) == ERR_EXTENT
    windowed_normals = img(abi.FMT_RG16_UNORM, 128, 64, full=(128, fh), origin=(0, 32))
    assert call(WindowPush(1.0, 32, 96), n=windowed_normals) == ERR_EXTENT  # `normal` must be the whole-frame image
    assert "whole-frame" in (lib.vkr_last_error() or b"").decode()
    assert call(WindowPush(1.0, 32, 96), d=img(abi.FMT_RGBA32_SFLOAT, w2, h2)) == ERR_EXTENT
    assert call(WindowPush(1.0, 32, 96), m=img(abi.FMT_RG16_UNORM, w2, h2)) == ERR_FORMAT
    assert lib.vkr_sssr_validate(C.byref(rays), C.byref(mask), C.byref(data), C.byref(normal), None, None) == ERR_NULL
    assert lib.vkr_sssr_validate(C.byref(rays), C.byref(mask), C.byref(img(abi.FMT_RGBA32_SFLOAT, w2, h2)), C.byref(normal), C.byref(params), None) == ERR_EXTENT


def test_comm_entries_refuse_null(lib):
    lib.vkr_comm_selfcheck.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vkr_comm_selfcheck_bytes.argtypes = [C.c_int]
    lib.vkr_comm_selfcheck_bytes.restype = C.c_uint64
    assert lib.vkr_comm_selfcheck(None, None, None) == ERR_NULL
    assert lib.vkr_comm_selfcheck_bytes(8) > lib.vkr_comm_selfcheck_bytes(2) > 0
    before = lib.vkr_get_switches()
    lib.vkr_set_switches(abi.SWITCH_BLUR_NO_SKIP | abi.SWITCH_TAA_GENERIC)
    assert lib.vkr_get_switches() == abi.SWITCH_BLUR_NO_SKIP | abi.SWITCH_TAA_GENERIC
    lib.vkr_set_switches(before)


def test_emulated_communicator_rules(lib):
    """vkr_comm_create_emulated: argument rules, rank / world read back, the self check refuses it (it moves no bytes), and
    destroying it needs no RCCL.  (No device call: the exchanges themselves are covered on the GPU.)"""
    h = C.c_void_p(0)
    assert lib.vkr_comm_create_emulated(0, 4, 60.0, 15.0, None) == ERR_NULL
    for rank, world, gbps, us in ((4, 4, 60.0, 15.0), (-1, 4, 60.0, 15.0), (0, 0, 60.0, 15.0), (0, 4, 0.0, 15.0), (0, 4, 60.0, -1.0)):
        assert lib.vkr_comm_create_emulated(rank, world, gbps, us, C.byref(h)) == ERR_EXTENT, (rank, world, gbps, us)
        assert b"comm_create_emulated" in lib.vkr_last_error()
    assert lib.vkr_comm_create_emulated(2, 4, 60.0, 15.0, C.byref(h)) == 0 and h.value
    lib.vkr_comm_rank.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    r, w = C.c_int(-1), C.c_int(-1)
    assert lib.vkr_comm_rank(h, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (2, 4)
    lib.vkr_comm_selfcheck.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    scratch = (C.c_uint8 * 16)()
    assert lib.vkr_comm_selfcheck(h, scratch, None) != 0 and b"emulated" in lib.vkr_last_error()
    assert lib.vkr_comm_destroy(h) == 0
