"""Known-answer tests that pin the oracle.  The reference ships no tests,
goldens or runnable CPU path, so these analytic identities are what stands between the restatement
and a silent transcription error."""
import ctypes as C
import math

import numpy as np
import pytest

from vk_renderer_amd import abi
from vk_renderer_amd.camera import FOVY, ZFAR, ZNEAR, FrameSetup
from vk_renderer_amd.chain import PostFxChain
from vk_renderer_amd.images import ImageBuf

def _olib():
    from oracle import binding

    return binding.load()


F3 = C.c_float * 3
F2 = C.c_float * 2


def _typed(l):
    l.vkr_ref_linearize_depth2.restype = C.c_float
    l.vkr_ref_linearize_depth2.argtypes = [C.c_float] * 3
    l.vkr_ref_encode_depth.restype = C.c_float
    l.vkr_ref_encode_depth.argtypes = [C.c_float] * 3
    l.vkr_ref_gtao_direction.restype = C.c_float
    l.vkr_ref_gtao_direction.argtypes = [C.c_int, C.c_int]
    l.vkr_ref_reconstruct_view_vec.argtypes = [F2, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, F3]
    l.vkr_ref_project_view_vec.argtypes = [F3, C.c_float, C.c_float, C.c_float, C.c_float, F3]
    l.vkr_ref_encode_normal.argtypes = [F3, F2]
    l.vkr_ref_decode_normal.argtypes = [F2, F3]
    return l


@pytest.fixture(scope="module")
def lib(oracle_lib):
    return _typed(oracle_lib)


# (1) depth / projection round trips ---------------------------------------------------------------
def test_depth_roundtrip(lib):
    for z in (-0.05, -0.06, -0.5, -1.0, -3.7, -12.0, -79.9, -80.0):
        d = lib.vkr_ref_encode_depth(z, ZNEAR, ZFAR)
        assert 0.0 <= d <= 1.0 + 1e-6
        back = lib.vkr_ref_linearize_depth2(d, ZNEAR, ZFAR)
        assert back == pytest.approx(z, rel=2e-3 if z < -40 else 2e-4)  # fp32 cancellation grows towards zfar
    assert lib.vkr_ref_linearize_depth2(0.0, ZNEAR, ZFAR) == pytest.approx(-ZNEAR)
    assert lib.vkr_ref_linearize_depth2(1.0, ZNEAR, ZFAR) == pytest.approx(-ZFAR, rel=2e-4)  # fp32: d*(f-n) - f cancels


def test_project_reconstruct_roundtrip(lib):
    aspect = 16.0 / 9.0
    rng = np.random.default_rng(7)
    for _ in range(200):
        u, v = rng.uniform(0.02, 0.98, 2)
        d = float(rng.uniform(0.2, 0.999))
        out = F3()
        lib.vkr_ref_reconstruct_view_vec(F2(u, v), d, FOVY, aspect, ZNEAR, ZFAR, out)
        assert out[2] < 0  # view-space z is negative (gbuffer_encode.glsl:53-56)
        back = F3()
        lib.vkr_ref_project_view_vec(F3(*out), FOVY, aspect, ZNEAR, ZFAR, back)
        assert back[0] == pytest.approx(u, abs=2e-6) and back[1] == pytest.approx(v, abs=2e-6)
        assert back[2] == pytest.approx(d, abs=2e-6)


# (2) octahedral normals -----------------------------------------------------------------------------
def test_normal_roundtrip_including_negative_z(lib):
    rng = np.random.default_rng(3)
    vecs = rng.normal(size=(500, 3))
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    vecs = np.vstack([vecs, [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, -1, 0], [0.6, 0, -0.8]]])
    for n in vecs:
        e = F2()
        lib.vkr_ref_encode_normal(F3(*[float(v) for v in n]), e)
        assert 0.0 <= e[0] <= 1.0 and 0.0 <= e[1] <= 1.0
        d = F3()
        lib.vkr_ref_decode_normal(e, d)
        assert np.allclose([d[0], d[1], d[2]], n, atol=3e-6)


# (3) Halton(2,3) --------------------------------------------------------------------------------------
def test_halton_first_elements(lib):
    buf = (C.c_float * (4 * 128))()
    lib.vkr_ref_halton23(buf, 128)
    h = np.frombuffer(buf, dtype=np.float32).reshape(128, 4)
    want = [(1 / 2, 1 / 3), (1 / 4, 2 / 3), (3 / 4, 1 / 9), (1 / 8, 4 / 9)]
    for i, (a, b) in enumerate(want):
        assert h[i, 0] == pytest.approx(a, rel=1e-6) and h[i, 1] == pytest.approx(b, rel=1e-6)
    assert np.all(h[:, 2:] == 0) and np.all((h[:, :2] > 0) & (h[:, :2] < 1))
    # the Python chain and the C++ host build the same table
    c = PostFxChain(64, 32, backend="oracle")
    assert np.array_equal(c.halton_host, h)


# (4) slice direction pattern ---------------------------------------------------------------------------
def test_gtao_direction_table(lib):
    table = [[0, 5, 10, 15], [4, 9, 14, 3], [8, 13, 2, 7], [12, 1, 6, 11]]  # [y][x] * 16 (main.comp:276-278)
    for y in range(8):
        for x in range(8):
            assert lib.vkr_ref_gtao_direction(x, y) == table[y % 4][x % 4] / 16.0


# helpers --------------------------------------------------------------------------------------------------
def _plane_chain(w=64, h=32, z_view=-4.0, use_mis=0, static_camera=False):
    """G-buffer of a fronto-parallel plane at view depth z_view, normal facing the camera."""
    kw = dict(prev_delta=(0.0, 0.0, 0.0), prev_yaw_delta=0.0) if static_camera else {}
    setup = FrameSetup(w, h, use_mis=use_mis, **kw)
    c = PostFxChain(w, h, backend="oracle", setup=setup)
    _typed(_olib())
    d = float(_olib().vkr_ref_encode_depth(z_view, ZNEAR, ZFAR))
    d24 = np.uint32(round(d * 16777215.0))
    c.depth.set_raw(np.full((h, w, 1), d24, dtype=np.uint32))
    c.prev_depth.set_raw(np.full((h, w, 1), d24, dtype=np.uint32))
    # world normal such that the view-space normal is (0,0,1): n_world = view^-1 rotation * (0,0,1)
    n_world = setup.inv_view[:3, :3] @ np.array([0.0, 0.0, 1.0])
    e = F2()
    _olib().vkr_ref_encode_normal(F3(*[float(v) for v in n_world]), e)
    n16 = np.array([round(e[0] * 65535), round(e[1] * 65535)], dtype=np.uint16)
    c.normal.set_raw(np.broadcast_to(n16, (h, w, 2)).copy())
    mat = np.zeros((h, w, 4), dtype=np.uint8)
    mat[..., 1] = 188  # sRGB code of roughness ~0.5
    c.material.set_raw(mat)
    alb = np.full((h, w, 4), 128, dtype=np.uint8)
    c.albedo.set_raw(alb)
    c.velocity.set_raw(np.zeros((h, w, 2), dtype=np.float16))
    return c


# (5) flat plane: no occluders ------------------------------------------------------------------------------
def test_gtao_flat_plane_is_unoccluded():
    """Fronto-parallel plane: every horizon sample lies in the plane, so nothing occludes.  One slice
    direction per pixel (4x4 rotation pattern, main.comp:276-278) integrates only its own half-slice, whose
    value swings around 1 with the tilt between view ray and plane normal; the mean over a 4x4 pattern
    block is the full cosine-weighted visibility = 1, and at the screen centre (view ray = normal) every
    single direction gives 1: h = pi/2, n = 0 -> 2 * 0.25 * (-cos(pi) + cos(0)) = 1."""
    c = _plane_chain(256, 128)
    c.downsample()
    c.gtao_main()
    ao = c.raw.decode()[..., 0]
    inner = ao[8:-8, 16:-16]
    h, w = inner.shape
    blocks = inner.reshape(h // 4, 4, w // 4, 4).mean(axis=(1, 3))
    assert blocks.min() > 0.93 and blocks.max() < 1.03, (blocks.min(), blocks.max())
    cy, cx = ao.shape[0] // 2, ao.shape[1] // 2
    assert np.all(np.abs(ao[cy - 2:cy + 2, cx - 2:cx + 2] - 1.0) < 0.06)
    assert np.allclose(c.raw.decode()[..., 1], 1.0 / (2.0 * math.pi), rtol=1e-3)  # occlusion.y untouched in non-MIS mode


def test_gtao_sky_pixels():
    for mis, want in ((0, (0.0, 1.0 / (2.0 * math.pi))), (1, (0.0, 1.0))):
        c = _plane_chain(use_mis=mis)
        c.depth.set_raw(np.full((32, 64, 1), 0xFFFFFF, dtype=np.uint32))
        c.downsample()
        c.preintegrate_pdf() if mis else None
        c.gtao_main()
        out = c.raw.decode()
        assert np.allclose(out[..., 0], want[0]) and np.allclose(out[..., 1], want[1], rtol=1e-3)


# (6) Hi-Z ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [(64, 32), (66, 38), (100, 36), (2, 2), (130, 2)])
def test_hiz_every_mip_is_min_of_parents(size):
    w, h = size
    rng = np.random.default_rng(w * 1000 + h)
    from vk_renderer_amd.images import depth_mip_count

    L = depth_mip_count(w, h)
    depth = ImageBuf(abi.FMT_D24_UNORM_S8, w, h, L)
    depth.set_raw(rng.integers(0, 1 << 24, size=(h, w, 1), dtype=np.uint32))
    n = ImageBuf(abi.FMT_RG16_UNORM, w, h)
    n.set_raw(rng.integers(0, 1 << 16, size=(h, w, 2), dtype=np.uint16))
    v = ImageBuf(abi.FMT_RG16_SFLOAT, w, h)
    v.set_raw(rng.integers(0, 0x7BFF, size=(h, w, 2), dtype=np.uint16).view(np.float16))
    dn = ImageBuf(abi.FMT_RG16_UNORM, max(1, w // 2), max(1, h // 2))
    dv = ImageBuf(abi.FMT_RG16_SFLOAT, max(1, w // 2), max(1, h // 2))
    lib = _olib()
    assert lib.vkr_ref_downsample_gbuffer(C.byref(depth.desc()), C.byref(n.desc()), C.byref(v.desc()), C.byref(dn.desc()), C.byref(dv.desc())) == 0
    assert lib.vkr_ref_depth_mips(C.byref(depth.desc()), 1) == 0
    for m in range(1, L):
        p, q = depth.raw(m - 1)[..., 0] & 0xFFFFFF, depth.raw(m)[..., 0]
        ph, pw = p.shape
        for y in range(q.shape[0]):
            for x in range(q.shape[1]):
                vals = [int(p[yy, xx]) if (yy < ph and xx < pw) else 0 for yy in (2 * y, 2 * y + 1) for xx in (2 * x, 2 * x + 1)]
                assert int(q[y, x]) == min(vals), (m, x, y)  # odd extents drop the last row/column; OOB fetch = 0
    # downsampled normal / velocity come from the texel holding the minimum (first match d1, d2, d3 else d0)
    d0 = depth.raw(0)[..., 0] & 0xFFFFFF
    for y in range(dn.height):
        for x in range(dn.width):
            quad = [(0, 0), (1, 0), (0, 1), (1, 1)]
            vals = [int(d0[2 * y + oy, 2 * x + ox]) for ox, oy in quad]
            mn = min(vals)
            pick = next((k for k in (1, 2, 3) if vals[k] == mn), 0)
            ox, oy = quad[pick]
            assert np.array_equal(dn.raw(0)[y, x], n.raw(0)[2 * y + oy, 2 * x + ox])
            assert np.array_equal(dv.raw(0)[y, x].view(np.uint16), v.raw(0)[2 * y + oy, 2 * x + ox].view(np.uint16))


def test_hiz_rejects_single_mip_and_mismatched_outputs():
    lib = _olib()
    depth = ImageBuf(abi.FMT_D24_UNORM_S8, 16, 16, 1)
    n, v = ImageBuf(abi.FMT_RG16_UNORM, 16, 16), ImageBuf(abi.FMT_RG16_SFLOAT, 16, 16)
    dn, dv = ImageBuf(abi.FMT_RG16_UNORM, 8, 8), ImageBuf(abi.FMT_RG16_SFLOAT, 8, 8)
    assert lib.vkr_ref_downsample_gbuffer(C.byref(depth.desc()), C.byref(n.desc()), C.byref(v.desc()), C.byref(dn.desc()), C.byref(dv.desc())) != 0
    depth = ImageBuf(abi.FMT_D24_UNORM_S8, 16, 16, 3)
    bad = ImageBuf(abi.FMT_RG16_UNORM, 4, 8)
    assert lib.vkr_ref_downsample_gbuffer(C.byref(depth.desc()), C.byref(n.desc()), C.byref(v.desc()), C.byref(bad.desc()), C.byref(dv.desc())) != 0


# (7) PDF LUT vs quadrature -----------------------------------------------------------------------------------------
def test_pdf_lut_matches_quadrature():
    from scipy import integrate

    c = PostFxChain(64, 32, backend="oracle")
    c.preintegrate_pdf()
    lut = c.pdf.decode()[..., 0]

    def G2(t, a, b):
        L = (b - a) * t + (b + a)
        return (1 - t) * L / (1 + t * t - 0.5 * L * L) ** 2 if L > 0 else 0.0

    for (x, y) in [(512, 512), (100, 300), (900, 200), (512, 100), (300, 700)]:
        a, b = 2 * (x + 0.5) / 1024 - 1, (y + 0.5) / 1024
        val, _ = integrate.quad(G2, -1, 1, args=(a, b), limit=400, points=[-(b + a) / (b - a)] if abs(b - a) > 1e-9 and -1 < -(b + a) / (b - a) < 1 else None)
        assert lut[y, x] == pytest.approx(val, rel=5e-3), (x, y)


# (8) TAA ---------------------------------------------------------------------------------------------------------------
def test_taa_static_scene_blends_one_tenth():
    c = _plane_chain(use_mis=1, static_camera=True)
    hist = np.zeros((32, 64, 4), dtype=np.float16)
    hist[..., :3] = 0.25
    c.taa_hist.set_raw(hist)
    c.taa()
    out = c.taa_target.decode()
    cur = c.albedo.decode()[0, 0, 0]
    assert np.allclose(out[..., :3], 0.9 * 0.25 + 0.1 * cur, rtol=2e-3)
    assert np.all(out[..., 3] == 0)


def test_taa_history_outside_screen_returns_current():
    c = _plane_chain(use_mis=1)
    c.velocity.set_raw(np.full((32, 64, 2), 2.0, dtype=np.float16))  # prev_uv = uv + 2 -> outside [0,1]
    hist = np.zeros((32, 64, 4), dtype=np.float16)
    hist[..., :3] = 0.9
    c.taa_hist.set_raw(hist)
    c.taa()
    out = c.taa_target.decode()
    cur = c.albedo.decode()[..., :3]
    assert np.allclose(out[..., :3], cur.astype(np.float16).astype(np.float32), atol=1e-3)


# (9) accumulate ---------------------------------------------------------------------------------------------------------
def test_accumulate_clear_history_and_saturation():
    c = _plane_chain(use_mis=1, static_camera=True)
    c.downsample()
    c.build_prev_hiz()
    c.filtered.set_raw(np.full((16, 32, 1), 0.5, dtype=np.float16))
    hist = np.zeros((16, 32, 2), dtype=np.float16)
    hist[..., 0], hist[..., 1] = 0.25, 1.0  # 255 samples accumulated
    c.acc_hist.set_raw(hist)
    c.gtao_accumulate(clear_history=1)
    out = c.acc_ao.decode()
    assert np.allclose(out[..., 0], 0.5) and np.allclose(out[..., 1], 1.0 / 255.0, rtol=1e-3)
    c.gtao_accumulate(clear_history=0)
    out = c.acc_ao.decode()
    # n = 255 * 1.0 * valid(=1) -> ao = (0.25*255 + 0.5)/256, n+1 = 256 > 255 -> stored count 100/255
    assert np.allclose(out[..., 0], (0.25 * 255 + 0.5) / 256, rtol=2e-3)
    assert np.allclose(out[..., 1], 100.0 / 255.0, rtol=1e-3)


# (10) blur -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rough_code", [0, 90, 188, 255])
def test_blur_constant_input_is_identity(rough_code):
    c = _plane_chain(use_mis=1)
    mat = np.zeros((32, 64, 4), dtype=np.uint8)
    mat[..., 1] = rough_code
    c.material.set_raw(mat)
    c.downsample()
    c.build_prev_hiz()
    refl = np.zeros((16, 32, 4), dtype=np.uint8)
    refl[..., 0], refl[..., 1], refl[..., 2] = 200, 100, 50
    c.reflections.set_raw(refl)
    c.ssr_blur(accumulate=0)
    out = c.blurred.raw(0)
    r = int(math.floor(3 * (0.4 + 3.6 * c.material.decode()[0, 0, 1]) - 0.01))
    inner = out[r:-r or None, r:-r or None] if 2 * r < 16 else out[0:0]
    assert np.all(np.abs(inner[..., :3].astype(int) - np.array([200, 100, 50])) <= 1)


# (11) filter ---------------------------------------------------------------------------------------------------------------
def test_filter_all_rays_invalid_gives_zero():
    c = _plane_chain(use_mis=1)
    c.downsample()
    rays = np.zeros((16, 32, 4), dtype=np.uint16)
    rays[..., 0], rays[..., 1], rays[..., 2], rays[..., 3] = 30000, 30000, 60000, 65535  # w == 1.0 -> invalid
    c.rays.set_raw(rays)
    c.ssr_filter()
    out = c.reflections.raw(0)
    assert np.all(out[1:-1, 1:-1, :3] == 0)  # interior: every tap is an invalid ray (edge taps fetch out of bounds = valid garbage)


# (12) deferred shading ------------------------------------------------------------------------------------------------------
def test_shading_show_ao_writes_srgb_of_occlusion():
    """show_ao != 0: out = (occlusion, occlusion, occlusion) written to an sRGB attachment (shader.frag:96-97)."""
    c = _plane_chain(use_mis=1, static_camera=True)
    c.downsample()
    c.preintegrate_brdf()
    acc = np.zeros((16, 32, 2), dtype=np.float16)
    acc[..., 0] = 0.5
    c.acc_ao.set_raw(acc)
    c.blurred.set_raw(np.zeros((16, 32, 4), dtype=np.uint8))
    c.shading(show_ao=1)
    out = c.color_out.raw(0)
    assert np.all(out[..., :3] == 188) and np.all(out[..., 3] == 0)  # sRGB code of 0.5, alpha 0


def test_brdf_lut_is_a_split_sum():
    """A + B = mean(G2/G1) <= 1 and both non-negative; at roughness -> 0 and NdotV -> 1 the lobe is a mirror: A ~ 1, B ~ 0."""
    c = PostFxChain(64, 32, backend="oracle")
    c.preintegrate_brdf()
    lut = c.brdf.decode()
    assert np.all(lut[..., :2] >= -1e-3) and np.all(lut[..., 0] + lut[..., 1] <= 1.0 + 2e-3)
    assert lut[1020, 2, 0] == pytest.approx(1.0, abs=2e-2) and lut[1020, 2, 1] == pytest.approx(0.0, abs=2e-2)


# ---- rows G4 / R2: the passes the reference ships but never records -----------------------------------
def _variant_chain(w=128, h=64):
    c = PostFxChain(w, h, backend="oracle")
    c.synth()
    c.build_prev_hiz()
    c.downsample()
    return c


def test_static_reprojection_blend(oracle_lib):
    """reproject.comp / accumulate.comp: identical depth => mix(prev, new, 0.05); changed depth => new."""
    c = _variant_chain()
    c.setup.use_mis = 0
    c.gtao_main()
    c.gtao_filter()
    c._half_img(abi.FMT_R16_SFLOAT, "ao_prev_frame")
    c._half_img(abi.FMT_R16_SFLOAT, "ao_output")
    hist = c.ao_prev_frame.to_host()
    hist.view(np.uint16)[:] = 0x3800  # 0.5
    c.ao_prev_frame.upload(hist)
    c.prev_depth.upload(c.depth.to_host())  # static camera
    c.gtao_reproject()
    new = c.filtered.decode()[..., 0]
    out = c.ao_output.decode()[..., 0]
    sky = c.depth.decode(1)[..., 0] >= 1.0
    tw, th = (c.raw.width // 8) * 8, (c.raw.height // 4) * 4
    exp = np.where(sky, new, np.float32(0.5) * np.float32(0.95) + new * np.float32(0.05))
    assert np.allclose(out[:th, :tw], exp[:th, :tw], rtol=2e-3, atol=1e-3)
    assert (~sky).any() and sky.any()


def test_deinterleave_layer_mapping(oracle_lib):
    """deinterleave.comp: texel (x, y) -> layer ((y&3)<<2)+(x&3) at (x>>2, y>>2); dispatch covers the layer extent only."""
    c = _variant_chain(256, 128)
    c._layer_descs(2)
    c.deinterleave_depth(2)
    src = c.depth.raw(1)[..., 0] & 0xFFFFFF
    lw, lh = c.deint_layers[0].width, c.deint_layers[0].height
    tw, th = (lw // 8) * 8, (lh // 4) * 4
    for y in range(0, th, 3):
        for x in range(0, tw, 5):
            layer = ((y & 3) << 2) + (x & 3)
            got = c.deint_layers[layer].decode()[y >> 2, x >> 2, 0]
            assert got == np.float32(np.float64(src[y, x]) / 16777215.0)
    # nothing beyond the dispatched extent was written
    assert c.deint_layers[0].decode()[(th >> 2) + 1:, :, 0].max(initial=0.0) == 0.0


def test_screen_trace_filter_constant(oracle_lib):
    """screen_trace/filter.comp is a normalised weighted mean: constant raw in => the same constant out."""
    c = _variant_chain()
    c._full_img("st_raw")
    h = c.st_raw.to_host()
    h.view(np.uint16)[:] = 0x3400  # 0.25 in every channel
    c.st_raw.upload(h)
    c.screen_trace_filter()
    out = c.st_filtered.decode()
    tw, th = (c.st_raw.width // 8) * 8, (c.st_raw.height // 4) * 4
    inner = out[2:th - 1, 2:tw - 1]  # taps of edge texels fall outside the image and fetch 0
    assert np.allclose(inner, 0.25, rtol=2e-3)


def test_screen_trace_sky_and_range(oracle_lib):
    """trace.comp:232-235: sky => (0,0,0,1); elsewhere the AO term 2*0.25*(1-cos 2h) lies in [0, 1]."""
    c = _variant_chain(256, 144)
    c.screen_trace()
    raw = c.st_raw.decode()
    tw, th = (c.st_raw.width // 8) * 8, (c.st_raw.height // 8) * 8
    uvx = (np.arange(tw, dtype=np.float32) / np.float32(tw))
    # sky test uses the bilinear depth at uv = pos/size; restrict to texels whose 2x2 footprint is all sky
    d = c.depth.decode(0)[..., 0]
    sky4 = (d >= 1.0)
    sky4[1:, :] &= sky4[:-1, :]
    sky4[:, 1:] &= sky4[:, :-1]
    sel = sky4[:th, :tw]
    assert sel.any()
    assert np.all(raw[:th, :tw][sel] == np.array([0, 0, 0, 1], dtype=np.float32))
    a = raw[:th, :tw, 3]
    assert a.min() >= 0.0 and a.max() <= 1.0 + 1e-3
    assert uvx[0] == 0.0


def test_tile_classification_partition(oracle_lib):
    """classification.comp: every 8x8 tile lands in exactly one list; the threshold moves tiles monotonically from
    glossy to reflective; glossy_value above max_roughness classifies everything as reflective."""
    c = _variant_chain(256, 144)
    w2, h2 = 128, 72
    total = ((w2 + 7) // 8) * ((h2 + 7) // 8)
    prev_reflective = -1
    for g in (0.0, 0.3, 0.5, 0.8, 1.01):
        c.ssr_classify(glossy_value=g)
        nr, ng = int(c.reflective_args[0]), int(c.glossy_args[0])
        assert list(c.reflective_args[1:]) == [1, 1] and list(c.glossy_args[1:]) == [1, 1]
        assert nr + ng == total
        both = np.concatenate([c.reflective_tiles[:nr], c.glossy_tiles[:ng]])
        assert np.array_equal(np.sort(both), np.arange(total))
        assert nr >= prev_reflective
        prev_reflective = nr
    assert prev_reflective == total
    c.ssr_classify(glossy_value=0.0)
    assert int(c.reflective_args[0]) == 0


# ---- the SSR trace against analytic geometry -----------------------------------------------------------------------------
def test_trace_mirror_floor_hits_wall_where_geometry_says(oracle_lib):
    """A perfectly smooth floor (y = 0) in front of a wall (z = 9): for roughness 0 the VNDF sample is the surface normal
    (brdf.glsl:135-155 degenerates), so the traced ray is the mirror reflection and must end on the wall where the law of
    reflection puts it — computed here in float64 from the camera matrices alone.  Pins reconstruct/project conventions,
    reflect(), the Hi-Z march and the validity tests of trace.comp:94-118 against geometry instead of against themselves."""
    from vk_renderer_amd import scene as scn

    W, H = 512, 288
    setup = FrameSetup(W, H)
    sc = scn.Scene()
    smooth = sc.add_texture(np.tile(np.array([128, 0, 0, 255], np.uint8), (4, 4, 1)))  # material.g = 0 -> roughness 0
    quad = sc.add_mesh(np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32), np.array([[0, 1, 0]] * 4, np.float32),
                       np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), np.array([0, 2, 1, 0, 3, 2], np.uint32))
    floor = np.array([[30, 0, 0, 0], [0, 1, 0, 0], [0, 0, 30, 9], [0, 0, 0, 1]], np.float32)       # y = 0, huge
    wall = np.array([[30, 0, 0, 0], [0, 0, -30, 0], [0, -1, 0, 9], [0, 0, 0, 1]], np.float32)       # z = 9, normal (0, 0, -1)
    sc.add_draw(sc.add_transform(floor), quad, scn.INVALID, smooth)
    sc.add_draw(sc.add_transform(wall), quad, scn.INVALID, smooth)
    c = PostFxChain(W, H, backend="oracle", setup=setup)
    c.raster(sc)
    c.downsample()
    c.preintegrate_pdf()
    c.ssr_trace(frame_random=0)
    rays = c.rays.decode()  # (uv.x, uv.y, depth, valid ? pixel_depth : 1)
    w2, h2 = W // 2, H // 2

    mvp = setup.mvp.astype(np.float64)
    inv = np.linalg.inv(mvp)
    checked = 0
    worst = 0.0
    for py in range(h2 // 2 + 8, h2 - 4, 7):
        for px in range(8, w2 - 8, 13):
            u, v = (px + 0.5) / w2, (py + 0.5) / h2
            a = inv @ np.array([2 * u - 1, 2 * v - 1, 0.0, 1.0])
            b = inv @ np.array([2 * u - 1, 2 * v - 1, 1.0, 1.0])
            a, b = a[:3] / a[3], b[:3] / b[3]
            d = (b - a) / np.linalg.norm(b - a)
            if abs(d[1]) < 1e-6:
                continue
            t = -a[1] / d[1]
            P = a + t * d                      # floor point seen through the pixel
            if t <= 0 or P[2] >= 9.0:
                continue                       # the pixel shows the wall (or nothing), not the floor
            r = d * np.array([1.0, -1.0, 1.0])  # mirror about y = 0
            s = (9.0 - P[2]) / r[2]
            Q = P + s * r                      # where the reflection meets the wall
            q = mvp @ np.append(Q, 1.0)
            hit_uv = (q[:2] / q[3]) * 0.5 + 0.5
            if not (0.02 < hit_uv[0] < 0.98 and 0.02 < hit_uv[1] < 0.98):
                continue
            got = rays[py, px]
            assert got[3] != 1.0, f"ray of pixel ({px}, {py}) should be a valid hit"
            err = max(abs(got[0] - hit_uv[0]) * w2, abs(got[1] - hit_uv[1]) * h2)
            worst = max(worst, err)
            assert err <= 2.0, f"pixel ({px}, {py}): hit {got[:2]} vs geometric {hit_uv} ({err:.2f} half-res px)"
            assert abs(got[2] - q[2] / q[3]) < 2e-3  # depth of the hit point
            checked += 1
    assert checked > 40, f"only {checked} mirror pixels checked: the scene does not exercise the trace"
    print(f"[known-answer] mirror trace: {checked} pixels, worst {worst:.2f} half-res px")


def test_velocity_convention_reprojects_to_the_same_surface_point(oracle_lib):
    """opaque_taa.frag:45 writes velocity = 0.5 (ndc_prev - ndc_cur) and resolve.comp:27-31 fetches the history at
    uv + velocity.  For a static textured scene seen from two cameras, the history (= the previous camera's image)
    fetched that way must show the same surface point as the current pixel; with the sign flipped it must not."""
    from vk_renderer_amd import scene as scn
    from vk_renderer_amd.images import ImageBuf

    W, H = 384, 216
    setup = FrameSetup(W, H)
    sc = scn.procedural_scene(detail=16)
    cur = PostFxChain(W, H, backend="oracle", setup=setup)
    cur.raster(sc)
    cur.raster(sc, target="prev")

    class PrevSetup(FrameSetup):  # the previous camera as "current" camera of a second chain
        def __init__(self):
            super().__init__(W, H)
            self.mvp, self.view, self.inv_view = setup.prev_mvp, setup.prev_view, setup.prev_inv_view

    prev = PostFxChain(W, H, backend="oracle", setup=PrevSetup())
    prev.raster(sc)

    def run(sign):
        # history = previous camera's albedo as RGBA16F, colour = current albedo
        hist = np.zeros((H, W, 4), np.float16)
        hist[..., :3] = prev.albedo.decode()[..., :3]
        cur.taa_hist.set_raw(hist.view(np.uint16))
        vel = cur.velocity.raw(0).view(np.float16).astype(np.float32) * sign
        v = ImageBuf(abi.FMT_RG16_SFLOAT, W, H)
        v.set_raw(vel.astype(np.float16).view(np.uint16))
        keep = cur.velocity
        cur.velocity = v
        cur.taa()
        cur.velocity = keep
        out = cur.taa_target.decode()[..., :3]
        col = cur.albedo.decode()[..., :3]
        return float(np.abs(out - col).mean())

    good, bad = run(+1.0), run(-1.0)
    # the same statement without the resolve pass: previous image at uv + velocity vs the current image
    vel = cur.velocity.raw(0).view(np.float16).astype(np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    col, pimg = cur.albedo.decode()[..., :3], prev.albedo.decode()[..., :3]

    def direct(sign):
        px = np.clip(np.rint((xs + 0.5) + sign * vel[..., 0] * W - 0.5), 0, W - 1).astype(int)
        py = np.clip(np.rint((ys + 0.5) + sign * vel[..., 1] * H - 0.5), 0, H - 1).astype(int)
        return float(np.abs(pimg[py, px] - col).mean())

    dgood, dbad = direct(+1.0), direct(-1.0)
    print(f"[known-answer] mean |history - current|: resolve {good:.4f} (flipped {bad:.4f}), direct fetch {dgood:.4f} (flipped {dbad:.4f})")
    assert dgood < 0.5 * dbad, "raster velocity does not point at the previous position of the surface point"
    assert good < 0.6 * bad, "resolve.comp does not fetch its history at uv + velocity"
