"""G-buffer raster stage: scene loader, mip builder, the oracle rasterizer's known answers
(CPU) and HIP-vs-oracle parity (GPU)."""
import json
import os

import numpy as np
import pytest

from vk_renderer_amd import scene as scn
from vk_renderer_amd.camera import FrameSetup
from vk_renderer_amd.chain import PostFxChain

SUZANNE = "/root/reference/assets/gltf/suzanne/Suzanne.gltf"


def _write_gltf(tmp_path):
    """A two-node glTF written on the spot: a unit quad (uint16 indices, TRS node under a matrix node) with one texture."""
    from PIL import Image

    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    nrm = np.array([[0, 0, 1]] * 4, np.float32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    blob = idx.tobytes() + pos.tobytes() + nrm.tobytes() + uv.tobytes()
    (tmp_path / "q.bin").write_bytes(blob)
    img = np.zeros((8, 4, 4), np.uint8)
    img[..., 0], img[..., 1], img[..., 3] = 200, np.arange(8)[:, None] * 30, 255
    Image.fromarray(img, "RGBA").save(tmp_path / "t.png")
    g = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [{"matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 5, 0, 0, 1], "children": [1]},
                  {"mesh": 0, "translation": [0, 2, 0], "scale": [2, 2, 2], "rotation": [0, 0, 0.7071068, 0.7071068]}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 1, "NORMAL": 2, "TEXCOORD_0": 3}, "indices": 0, "material": 0}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}, "alphaMode": "MASK"}],
        "textures": [{"source": 0}], "images": [{"uri": "t.png"}],
        "buffers": [{"uri": "q.bin", "byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 12}, {"buffer": 0, "byteOffset": 12, "byteLength": 48},
                        {"buffer": 0, "byteOffset": 60, "byteLength": 48}, {"buffer": 0, "byteOffset": 108, "byteLength": 32}],
        "accessors": [{"bufferView": 0, "componentType": 5123, "count": 6, "type": "SCALAR"},
                      {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 3, "componentType": 5126, "count": 4, "type": "VEC2"}],
    }
    (tmp_path / "q.gltf").write_text(json.dumps(g))
    return tmp_path / "q.gltf"


def test_gltf_loader_subset(tmp_path):
    sc = scn.load_gltf(str(_write_gltf(tmp_path)))
    assert sc.vertices.shape == (4, 8) and list(sc.indices) == [0, 1, 2, 0, 2, 3]
    assert len(sc.draws) == 1 and sc.draws[0]["albedo"] == 0 and sc.draws[0]["mr"] == scn.INVALID and sc.draws[0]["flags"] == 0xFF
    model, normal = sc.transforms[0]
    # parent translate(5,0,0) * [translate(0,2,0) * rotZ(90 deg) * scale(2)]: x axis -> +y
    assert np.allclose(model @ np.array([1, 0, 0, 1], np.float32), [5, 4, 0, 1], atol=1e-5)
    assert np.allclose(normal, np.linalg.inv(model).T, atol=1e-6)
    levels = sc.textures[0]
    assert [lv.shape[:2] for lv in levels] == [(8, 4), (4, 2), (2, 1), (1, 1)]  # floor(log2(max)) + 1 levels


def test_mip_builder_is_linear_space_box():
    img = np.zeros((2, 2, 4), np.uint8)
    img[0, 0], img[0, 1], img[1, 0], img[1, 1] = (255, 0, 0, 255), (0, 0, 0, 255), (255, 0, 0, 0), (0, 0, 0, 0)
    top = scn.build_mips(img)[1][0, 0]
    assert top[0] == 188 and top[1] == 0 and top[3] == 128  # mean of linear 1,0,1,0 = 0.5 -> sRGB 188; alpha 0.5 -> rint(127.5) = 128
    const = np.full((16, 16, 4), 77, np.uint8)
    assert all((lv == 77).all() for lv in scn.build_mips(const))


@pytest.mark.skipif(not os.path.exists(SUZANNE), reason="reference asset not present on this machine")
def test_loads_reference_suzanne():
    sc = scn.load_gltf(SUZANNE)
    assert sc.vertices.shape == (11808, 8) and sc.indices.size == 11808 and len(sc.draws) == 1
    assert len(sc.textures) == 2 and sc.textures[0][0].shape == (1024, 1024, 4) and len(sc.textures[0]) == 11
    assert np.allclose(np.linalg.norm(sc.vertices[:, 3:6], axis=1), 1.0, atol=1e-3)


def _raster(chain, sc):
    chain.raster(sc)
    chain.sync()
    return {n: getattr(chain, n).raw(0).copy() for n in ("depth", "albedo", "normal", "material", "velocity")}


def _single_triangle_scene(verts_ndc, setup):
    """A triangle given directly in NDC: model = inverse(view_projection) would lose bits, so place it in clip space
    through an identity camera instead (the caller passes an identity-matrix FrameSetup stand-in)."""
    sc = scn.Scene()
    pos = np.array(verts_ndc, np.float32)
    mesh = sc.add_mesh(pos, np.array([[0, 0, 1]] * 3, np.float32), np.zeros((3, 2), np.float32), np.array([0, 1, 2], np.uint32))
    sc.add_draw(sc.add_transform(np.eye(4, dtype=np.float32)), mesh)
    return sc


class _IdentitySetup(FrameSetup):
    def __init__(self, w, h):
        super().__init__(w, h)
        self.mvp = np.eye(4, dtype=np.float32)
        self.prev_mvp = np.eye(4, dtype=np.float32)


def test_oracle_fill_rule_and_clear(oracle_lib):
    W, H = 16, 8
    c = PostFxChain(W, H, backend="oracle", setup=_IdentitySetup(W, H))
    # screen x = (ndc + 1) / 2 * 16: the square [4, 12) x [2, 6) as two triangles sharing the diagonal, z = 0.25
    def ndc(x, y):
        return [x / 8.0 - 1.0, y / 4.0 - 1.0, 0.25]
    sc = scn.Scene()
    pos = np.array([ndc(4, 2), ndc(12, 2), ndc(12, 6), ndc(4, 6)], np.float32)
    mesh = sc.add_mesh(pos, np.array([[0, 0, 1]] * 4, np.float32), np.zeros((4, 2), np.float32), np.array([0, 1, 2, 0, 2, 3], np.uint32))
    sc.add_draw(sc.add_transform(np.eye(4, dtype=np.float32)), mesh)
    out = _raster(c, sc)
    d = out["depth"][..., 0] & 0xFFFFFF
    covered = d != 0xFFFFFF
    want = np.zeros((H, W), bool)
    want[2:6, 4:12] = True  # pixel centres x + 0.5 in [4, 12): columns 4..11; no pixel drawn twice, none missed on the diagonal
    assert np.array_equal(covered, want)
    assert np.all(d[covered] == round(0.25 * 16777215))
    assert np.all(out["albedo"][~covered] == 0) and np.all(out["velocity"][~covered] == 0)
    a = out["albedo"][covered]
    assert np.all(a[:, :3] == 188) and np.all(a[:, 3] == 255)  # untextured: vec4(0.5, 0.5, 0.5, 1) -> sRGB 188
    assert np.all(out["material"][covered] == np.array([188, 243, 188, 128], np.uint8))  # (0.5, 0.9, 0.5, 0.5)


def test_oracle_depth_test_is_less_or_equal_and_near_clip(oracle_lib):
    W, H = 16, 8
    c = PostFxChain(W, H, backend="oracle", setup=_IdentitySetup(W, H))
    sc = scn.Scene()
    full = np.array([[-1, -1, 0.5], [3, -1, 0.5], [-1, 3, 0.5]], np.float32)  # covers the whole viewport at z = 0.5
    n0 = np.array([[0, 0, 1]] * 3, np.float32)
    n1 = np.array([[0, 1, 0]] * 3, np.float32)
    t = sc.add_transform(np.eye(4, dtype=np.float32))
    sc.add_draw(t, sc.add_mesh(full, n0, np.zeros((3, 2), np.float32), np.array([0, 1, 2], np.uint32)))
    sc.add_draw(t, sc.add_mesh(full, n1, np.zeros((3, 2), np.float32), np.array([0, 2, 1], np.uint32)))  # same depth, other winding, drawn later
    out = _raster(c, sc)
    enc = out["normal"]
    assert np.all(enc[..., 0] == 32768) and np.all(enc[..., 1] == 65535)  # the later draw (normal +y) won every tie: octahedral (0.5, 1.0)
    # a triangle crossing z = 0 is clipped, not dropped: its visible part keeps depths in [0, 1]
    sc2 = scn.Scene()
    tri = np.array([[-1, -1, -0.5], [1, -1, 0.5], [-1, 1, 0.5]], np.float32)
    sc2.add_draw(sc2.add_transform(np.eye(4, dtype=np.float32)), sc2.add_mesh(tri, n0, np.zeros((3, 2), np.float32), np.array([0, 1, 2], np.uint32)))
    d = _raster(c, sc2)["depth"][..., 0] & 0xFFFFFF
    hit = d != 0xFFFFFF
    assert hit.any() and not hit.all() and d[hit].min() < 0.1 * 16777215


def _cutout_kat_scene():
    """Front quad [4, 12) x [2, 6) at z = 0.25 textured 4x4 with alpha 0 in texel columns 0-1, in front of a
    full-viewport triangle at z = 0.5 (normal +y)."""
    def ndc(x, y, z):
        return [x / 8.0 - 1.0, y / 4.0 - 1.0, z]
    sc = scn.Scene()
    t = sc.add_transform(np.eye(4, dtype=np.float32))
    back = np.array([[-1, -1, 0.5], [3, -1, 0.5], [-1, 3, 0.5]], np.float32)
    sc.add_draw(t, sc.add_mesh(back, np.array([[0, 1, 0]] * 3, np.float32), np.zeros((3, 2), np.float32), np.array([0, 1, 2], np.uint32)))
    tex = np.zeros((4, 4, 4), np.uint8)
    tex[..., 0], tex[..., 3] = 255, 255
    tex[:, :2, 3] = 0
    pos = np.array([ndc(4, 2, 0.25), ndc(12, 2, 0.25), ndc(12, 6, 0.25), ndc(4, 6, 0.25)], np.float32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    mesh = sc.add_mesh(pos, np.array([[0, 0, 1]] * 4, np.float32), uv, np.array([0, 1, 2, 0, 2, 3], np.uint32))
    sc.add_draw(t, mesh, sc.add_texture(tex))
    return sc


def _check_cutout_kat(out):
    d = out["depth"][..., 0] & 0xFFFFFF
    front, back = round(0.25 * 16777215), round(0.5 * 16777215)
    # u = (x + 0.5 - 4) / 8 magnifies the 4 texels over 8 pixels; bilinear with REPEAT: the filtered alpha is exactly 0 only
    # where both taps lie in texel columns 0-1: pixels x = 5, 6 (x = 4 wraps to the opaque column 3, x = 7 touches column 2)
    want = np.full(d.shape, back)
    want[2:6, 4:12] = front
    want[2:6, 5:7] = back
    assert np.array_equal(d, want), "discarded fragments must leave the depth of the geometry behind"
    hole = out["normal"][2:6, 5:7]
    assert np.all(hole[..., 0] == 32768) and np.all(hole[..., 1] == 65535), "the back triangle's normal (+y) shows through the hole"
    assert np.all(out["albedo"][2:6, 5:7, :3] == 188), "... and its untextured albedo"
    kept = out["albedo"][2:6, 7:12]
    assert np.all(kept[..., 0] == 255) and np.all(kept[..., 3] > 0)


def test_oracle_alpha_discard_shows_what_is_behind(oracle_lib):
    """opaque_taa.frag:32-34: `if (out_albedo.a == 0) discard;` — the fragment writes neither depth nor colour."""
    W, H = 16, 8
    c = PostFxChain(W, H, backend="oracle", setup=_IdentitySetup(W, H))
    _check_cutout_kat(_raster(c, _cutout_kat_scene()))


@pytest.mark.gpu
def test_alpha_discard_known_answer_gpu(oracle_lib):
    """The same known answer on the HIP rasterizer (the discard is evaluated at coverage time, before the atomicMin),
    with and without the caller's opaque hint withheld."""
    W, H = 16, 8
    c = PostFxChain(W, H, backend="product", device="cuda", setup=_IdentitySetup(W, H))
    _check_cutout_kat(_raster(c, _cutout_kat_scene()))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(640, 360)])
def test_raster_parity_cutout(size, oracle_lib):
    """The procedural scene with a fence whose texture has alpha-0 holes, minified and magnified: coverage / depth must
    stay bit-exact against the oracle (a wrong discard shows as a depth mismatch), and a fair share of the fence must
    actually have been discarded."""
    W, H = size
    ref = PostFxChain(W, H, backend="oracle")
    gpu = PostFxChain(W, H, backend="product", device="cuda")
    solid, cut = scn.procedural_scene(), scn.procedural_scene(cutout=True)
    ref.raster(solid)
    d_solid = ref.depth.raw(0)[..., 0] & 0xFFFFFF
    for c in (ref, gpu):
        c.raster(cut)
    gpu.sync()
    a, b = gpu.depth.raw(0)[..., 0] & 0xFFFFFF, ref.depth.raw(0)[..., 0] & 0xFFFFFF
    assert np.array_equal(a, b), f"depth: {int((a != b).sum())} texels differ"
    fence = b != d_solid
    print(f"[parity] cutout: fence covers {int(fence.sum())} px")
    assert fence.sum() > 0.01 * W * H
    # the fence's bounding box holds both fence fragments and holes showing the solid scene
    ys, xs = np.nonzero(fence)
    box = (slice(ys.min(), ys.max() + 1), slice(xs.min(), xs.max() + 1))
    holes = (~fence[box]).mean()
    print(f"[parity] cutout: holes inside the fence's box {holes:.3f}")
    assert 0.15 < holes < 0.6
    for name in ("albedo", "normal", "material", "velocity"):
        r, g = getattr(ref, name), getattr(gpu, name)
        n = int((r.raw(0) != g.raw(0)).any(axis=-1).sum())
        assert n <= 1e-4 * W * H, f"{name}: {n} texels differ"


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(256, 144), (640, 360)])
def test_raster_parity(size, oracle_lib):
    """HIP visibility-buffer rasterizer vs the oracle's immediate-mode z-buffer: coverage / depth bit-exact, the
    shaded attachments within tolerance; then the rasterised G-buffer drives the whole chain on both sides."""
    from parity import report

    W, H = size
    sc = scn.procedural_scene()
    ref = PostFxChain(W, H, backend="oracle")
    gpu = PostFxChain(W, H, backend="product", device="cuda")
    for c in (ref, gpu):
        c.raster(sc)
        c.raster(sc, target="prev")
    gpu.sync()
    for name in ("depth", "prev_depth"):
        a, b = getattr(gpu, name).raw(0)[..., 0] & 0xFFFFFF, getattr(ref, name).raw(0)[..., 0] & 0xFFFFFF
        assert np.array_equal(a, b), f"{name}: {int((a != b).sum())} texels differ (coverage / depth must be bit-exact)"
    cov = (ref.depth.raw(0)[..., 0] & 0xFFFFFF) != 0xFFFFFF
    print(f"[parity] raster coverage {cov.mean():.3f}")
    assert 0.5 < cov.mean() < 1.0
    for name in ("albedo", "normal", "material", "velocity"):
        r, g = getattr(ref, name), getattr(gpu, name)
        n, _ = report(name, r.format, g.decode(), r.decode())
        assert n <= 1e-4 * W * H, f"{name}: {n} texels outside tolerance"
    # the rasterised G-buffer feeds the chain
    for name in ("albedo", "normal", "material", "velocity"):
        getattr(gpu, name).copy_from(getattr(ref, name))
    for c in (ref, gpu):
        c.build_prev_hiz()
        c.preintegrate_pdf()
        c.init_histories()
        c.frame()
    gpu.sync()
    for name in ("rays", "reflections", "blurred", "acc_ao", "taa_target"):
        r, g = getattr(ref, name), getattr(gpu, name)
        n, _ = report(name, r.format, g.decode(), r.decode())
        assert n <= 1e-4 * r.width * r.height, f"{name}: {n} texels outside tolerance"


@pytest.mark.gpu
def test_host_mirror_scene_renderer(oracle_lib):
    """scene::CompiledScene + SceneRenderer::draw_taa through the C++ rendergraph mirror (clear, vertex / index buffers,
    transform SSBO, bindless textures, one draw_indexed per primitive) against the oracle rasterizer, then a frame of
    the chain on the rasterised G-buffer."""
    from vk_renderer_amd import host
    from parity import mismatches

    W, H = 512, 288
    setup = FrameSetup(W, H)
    sc = scn.procedural_scene()
    frame = host.HostFrame(setup, device="cuda")
    frame.load_scene(sc)
    frame.run(host.STAGE_LUT)
    # the previous frame: rasterise from the previous camera, build its Hi-Z, swap it into prev_depth (main.cpp:416)
    frame.set_camera(setup.prev_view, setup.prev_view, setup.proj, setup.fazz)
    frame.run(host.STAGE_RASTER | host.STAGE_DOWNSAMPLE)
    frame.end_frame(swap_depth=True)
    frame.set_camera(setup.view, setup.prev_view, setup.proj, setup.fazz)
    frame.run(host.STAGE_RASTER)
    assert frame.last_tasks() == ["GbufferPass"]

    ref = PostFxChain(W, H, backend="oracle", setup=setup)
    ref.raster(sc)
    ref.raster(sc, target="prev")
    ref.build_prev_hiz()
    for name, exact in (("depth", True), ("prev_depth", True), ("albedo", False), ("normal", False), ("material", False), ("velocity", False)):
        got, want = frame.download(name), getattr(ref, name)
        if exact:
            for mip in range(want.mips if name == "prev_depth" else 1):
                assert np.array_equal(got.raw(mip)[..., 0] & 0xFFFFFF, want.raw(mip)[..., 0] & 0xFFFFFF), f"{name} mip {mip}"
        else:
            bad = int(mismatches(want.format, got.decode(), want.decode()).sum())
            print(f"[parity] host raster {name:10s} outside-tol {bad}")
            assert bad <= 1e-4 * W * H
    frame.close()
