"""Scene data for the G-buffer raster stage: a glTF 2.0 loader following what
src/scene/scene.cpp keeps of a file (one interleaved {pos, norm, uv} vertex buffer, one uint32 index
buffer, per-primitive offsets, base-colour / metallic-roughness texture indices, node transforms
composed as translate * rotate * scale), the mip chains scene/images.cpp builds for every texture,
the draw list SceneRenderer::update_scene derives (scene_renderer.cpp:105-131), and a procedural scene
for machines that hold no asset (the GPU box: the reference's assets do not travel).

Plain numpy on the host; `upload()` turns it into the ctypes vkr_raster_scene of include/vkr_postfx.h.
"""
import ctypes as C
import json
import math
import os

import numpy as np

from . import abi
from .images import ImageBuf

INVALID = 0xFFFFFFFF


def _srgb_tables():
    c = np.arange(256, dtype=np.float64) / 255.0
    dec = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)
    m = (np.arange(1, 256, dtype=np.float64) - 0.5) / 255.0
    thr = np.where(m <= 0.04045, m / 12.92, ((m + 0.055) / 1.055) ** 2.4).astype(np.float32)
    return dec, thr


_DEC, _THR = _srgb_tables()


def encode_srgb8(lin):
    """float32 linear -> sRGB8 code with the library's rule: largest code whose threshold is <= x."""
    return np.searchsorted(_THR, lin.astype(np.float32), side="right").astype(np.uint8)


def build_mips(rgba8):
    """scene/images.cpp:32-49,93-160: floor(log2(max(w, h))) + 1 levels, each a linear-filter blit of the
    previous one.  Frozen as: decode sRGB (alpha stays linear), average the 2x2 source block in fp32
    as (a + b) + (c + d) times 0.25, encode.  Odd extents clamp the second sample."""
    levels = [np.ascontiguousarray(rgba8, dtype=np.uint8)]
    h, w = rgba8.shape[:2]
    n = int(math.floor(math.log2(max(w, h)))) + 1
    for _ in range(1, n):
        src = levels[-1]
        sh, sw = src.shape[:2]
        dh, dw = max(1, sh // 2), max(1, sw // 2)
        lin = np.empty(src.shape, dtype=np.float32)
        lin[..., :3] = _DEC[src[..., :3]]
        lin[..., 3] = src[..., 3].astype(np.float32) / np.float32(255.0)
        y0 = np.minimum(2 * np.arange(dh), sh - 1)
        y1 = np.minimum(2 * np.arange(dh) + 1, sh - 1)
        x0 = np.minimum(2 * np.arange(dw), sw - 1)
        x1 = np.minimum(2 * np.arange(dw) + 1, sw - 1)
        a, b = lin[y0][:, x0], lin[y0][:, x1]
        c, d = lin[y1][:, x0], lin[y1][:, x1]
        avg = ((a + b) + (c + d)) * np.float32(0.25)
        out = np.empty((dh, dw, 4), dtype=np.uint8)
        out[..., :3] = encode_srgb8(avg[..., :3])
        out[..., 3] = np.rint(np.clip(avg[..., 3], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
        levels.append(out)
    return levels


class Scene:
    """Host-side scene: vertices [N, 8] f32 (pos, norm, uv), indices [M] u32, transforms [(model, normal)] as 4x4
    maths-convention arrays, draws [dict(transform, albedo, mr, flags, index_offset, index_count, vertex_offset)],
    textures [list of mip levels (h, w, 4) u8]."""

    def __init__(self):
        self.vertices = np.zeros((0, 8), dtype=np.float32)
        self.indices = np.zeros((0,), dtype=np.uint32)
        self.transforms = []
        self.draws = []
        self.textures = []

    # ---- construction helpers ---------------------------------------------------------------------
    def add_mesh(self, pos, norm, uv, idx):
        v0, i0 = len(self.vertices), len(self.indices)
        v = np.concatenate([pos, norm, uv], axis=1).astype(np.float32)
        self.vertices = np.concatenate([self.vertices, v])
        self.indices = np.concatenate([self.indices, np.asarray(idx, dtype=np.uint32).reshape(-1)])
        return v0, i0, int(np.asarray(idx).size)

    def add_transform(self, model):
        model = np.asarray(model, dtype=np.float32)
        normal = np.linalg.inv(model.astype(np.float64)).T.astype(np.float32)  # transpose(inverse(model)), scene_renderer.cpp:111
        self.transforms.append((model, normal))
        return len(self.transforms) - 1

    def add_texture(self, rgba8):
        self.textures.append(build_mips(rgba8))
        return len(self.textures) - 1

    def add_draw(self, transform, mesh, albedo=INVALID, mr=INVALID, flags=0):
        v0, i0, n = mesh
        self.draws.append(dict(transform=transform, albedo=albedo, mr=mr, flags=flags, index_offset=i0, index_count=n, vertex_offset=v0))

    # ---- C-ABI view -------------------------------------------------------------------------------------
    def upload(self, device=None):
        """-> (abi.RasterScene, keep-alive list).  device None: host memory, else torch device."""
        keep = []

        def dev(arr):
            arr = np.ascontiguousarray(arr)
            if device is None:
                keep.append(arr)
                return arr.ctypes.data
            import torch

            t = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(device)
            keep.append(t)
            return t.data_ptr()

        s = abi.RasterScene()
        s.vertices, s.vertex_count = dev(self.vertices), len(self.vertices)
        s.indices, s.index_count = dev(self.indices), len(self.indices)
        tr = (abi.RasterTransform * max(1, len(self.transforms)))()
        for i, (m, n) in enumerate(self.transforms):
            tr[i].model, tr[i].normal = abi.Mat4.from_np(m), abi.Mat4.from_np(n)
        dr = (abi.RasterDraw * max(1, len(self.draws)))()
        # VKR_RASTER_DRAW_OPAQUE_ALBEDO: no texel of any level of the albedo texture has alpha 0, so the discard of
        # opaque_taa.frag:32-34 cannot fire and the stage may skip the coverage-time alpha test
        opaque = [all(int(lv[..., 3].min()) > 0 for lv in levels) for levels in self.textures]
        for i, d in enumerate(self.draws):
            hint = 1 if (d["albedo"] != INVALID and opaque[d["albedo"]] and not d.get("force_alpha_test")) else 0
            dr[i] = abi.RasterDraw(d["transform"], d["albedo"], d["mr"], d["flags"], d["index_offset"], d["index_count"], d["vertex_offset"], hint)
        tx = (abi.VkrImg * max(1, len(self.textures)))()
        for i, levels in enumerate(self.textures):
            h, w = levels[0].shape[:2]
            img = ImageBuf(abi.FMT_RGBA8_SRGB, w, h, len(levels))
            for m, lv in enumerate(levels):
                img.set_raw(lv, m)
            if device is not None:
                gpu_img = ImageBuf(abi.FMT_RGBA8_SRGB, w, h, len(levels), device=device)
                gpu_img.upload(img.to_host())
                img = gpu_img
            keep.append(img)
            tx[i] = img.desc()
        s.transforms, s.transform_count = C.cast(tr, C.c_void_p), len(self.transforms)
        s.draws, s.draw_count = C.cast(dr, C.c_void_p), len(self.draws)
        s.textures, s.texture_count = C.cast(tx, C.c_void_p), len(self.textures)
        keep += [tr, dr, tx]
        return s, keep


# ---- glTF ------------------------------------------------------------------------------------------------
_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_TYPE_N = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _quat_to_mat(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0],
                     [0, 0, 0, 1]], dtype=np.float64)


def load_gltf(path):
    """The subset scene.cpp reads: meshes / primitives (POSITION, NORMAL, TEXCOORD_0, indices), materials'
    baseColorTexture / metallicRoughnessTexture, the default scene's node tree (TRS or matrix)."""
    from PIL import Image

    folder = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        g = json.load(f)
    buffers = []
    for b in g.get("buffers", []):
        with open(os.path.join(folder, b["uri"]), "rb") as f:
            buffers.append(f.read())

    def accessor(i):
        a = g["accessors"][i]
        view = g["bufferViews"][a["bufferView"]]
        dt, n = np.dtype(_COMPONENT[a["componentType"]]), _TYPE_N[a["type"]]
        off = view.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = view.get("byteStride", 0) or dt.itemsize * n
        raw = np.frombuffer(buffers[view["buffer"]], dtype=np.uint8, count=stride * (a["count"] - 1) + dt.itemsize * n, offset=off)
        rows = np.lib.stride_tricks.as_strided(raw, shape=(a["count"], dt.itemsize * n), strides=(stride, 1))
        return np.ascontiguousarray(rows).view(dt).reshape(a["count"], n)

    sc = Scene()
    # textures: one image per glTF texture (scene.cpp:144-168); every image is loaded as RGBA8_SRGB (images.cpp:38)
    for t in g.get("textures", []):
        img = g["images"][t["source"]]
        sc.add_texture(np.array(Image.open(os.path.join(folder, img["uri"])).convert("RGBA")))
    materials = []
    for m in g.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        albedo = pbr.get("baseColorTexture", {}).get("index", -1)
        mr = pbr.get("metallicRoughnessTexture", {}).get("index", -1)
        ntex = len(sc.textures)
        materials.append(dict(albedo=albedo if 0 <= albedo < ntex else INVALID, mr=mr if 0 <= mr < ntex else INVALID,
                              flags=0xFF if m.get("alphaMode") == "MASK" else 0))
    meshes = []
    for m in g.get("meshes", []):
        prims = []
        for p in m["primitives"]:
            at = p["attributes"]
            if "POSITION" not in at:
                raise RuntimeError("No position")  # scene.cpp:206
            pos = accessor(at["POSITION"]).astype(np.float32)
            norm = accessor(at["NORMAL"]).astype(np.float32) if "NORMAL" in at else np.zeros_like(pos)
            uv = accessor(at["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in at else np.zeros((len(pos), 2), np.float32)
            idx = accessor(p["indices"]).astype(np.uint32).reshape(-1)
            prims.append((sc.add_mesh(pos, norm, uv, idx), p.get("material", -1)))
        meshes.append(prims)

    def node_matrix(n):
        if "matrix" in n:
            return np.array(n["matrix"], dtype=np.float64).reshape(4, 4).T  # glTF stores column-major
        m = np.eye(4)
        if "translation" in n:
            t = np.eye(4)
            t[:3, 3] = n["translation"]
            m = m @ t
        if "rotation" in n:
            m = m @ _quat_to_mat(n["rotation"])
        if "scale" in n:
            m = m @ np.diag(list(n["scale"]) + [1.0])
        return m

    def walk(idx, acc):  # scene_renderer.cpp:105-120
        n = g["nodes"][idx]
        m = acc @ node_matrix(n)
        if n.get("mesh", -1) >= 0:
            tid = sc.add_transform(m.astype(np.float32))
            for mesh, mat in meshes[n["mesh"]]:
                md = materials[mat] if 0 <= mat < len(materials) else dict(albedo=INVALID, mr=INVALID, flags=0)
                sc.add_draw(tid, mesh, md["albedo"], md["mr"], md["flags"])
        for c in n.get("children", []):
            walk(c, m)

    for root in g["scenes"][max(0, g.get("scene", 0))]["nodes"]:
        walk(root, np.eye(4))
    return sc


# ---- procedural stand-in -------------------------------------------------------------------------------------
def _uv_sphere(rings, segments, radius):
    pos, norm, uv, idx = [], [], [], []
    for r in range(rings + 1):
        th = math.pi * r / rings
        for s in range(segments + 1):
            ph = 2.0 * math.pi * s / segments
            n = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
            norm.append(n)
            pos.append(tuple(radius * c for c in n))
            uv.append((s / segments * 2.0, r / rings))
    for r in range(rings):
        for s in range(segments):
            a, b = r * (segments + 1) + s, (r + 1) * (segments + 1) + s
            idx += [a, b, a + 1, a + 1, b, b + 1]
    return np.array(pos, np.float32), np.array(norm, np.float32), np.array(uv, np.float32), np.array(idx, np.uint32)


def _checker(size, cells, c0, c1, seed):
    y, x = np.mgrid[0:size, 0:size]
    k = ((x * cells // size) + (y * cells // size)) & 1
    rng = np.random.default_rng(seed)
    noise = rng.integers(-12, 13, size=(size, size, 3))
    img = np.where(k[..., None] == 0, np.array(c0)[None, None, :], np.array(c1)[None, None, :]) + noise
    out = np.empty((size, size, 4), dtype=np.uint8)
    out[..., :3] = np.clip(img, 0, 255)
    out[..., 3] = 255
    return out


def procedural_scene(detail=24, cutout=False):
    """Ground quad, a back wall and three textured spheres in front of the reference camera (eye (0, 1, -1) looking
    along +z): exercises near/far depth ranges, both windings, shared edges, texture minification and magnification.
    cutout: adds a fence (a quad stood up at z = 5 whose texture has alpha-0 holes, like Sponza's foliage and chains):
    opaque_taa.frag:32-34 discards those fragments and the geometry behind shows through."""
    sc = Scene()
    albedo = sc.add_texture(_checker(256, 16, (200, 60, 50), (230, 220, 200), 1))
    mr = sc.add_texture(_checker(128, 8, (128, 70, 20), (128, 200, 230), 2))
    stone = sc.add_texture(_checker(256, 32, (90, 90, 100), (140, 140, 150), 3))
    sphere = sc.add_mesh(*_uv_sphere(detail, 2 * detail, 1.0))
    quad = sc.add_mesh(np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32), np.array([[0, 1, 0]] * 4, np.float32),
                       np.array([[0, 0], [8, 0], [8, 8], [0, 8]], np.float32), np.array([0, 2, 1, 0, 3, 2], np.uint32))

    def trs(t, s, rot_y=0.0):
        c, sn = math.cos(rot_y), math.sin(rot_y)
        m = np.array([[c * s[0], 0, sn * s[2], t[0]], [0, s[1], 0, t[1]], [-sn * s[0], 0, c * s[2], t[2]], [0, 0, 0, 1]], dtype=np.float32)
        return m

    sc.add_draw(sc.add_transform(trs((0, 0, 8), (14, 1, 14))), quad, stone, mr)                      # ground
    wall = np.array([[12, 0, 0, 0], [0, 0, -7, 3.5], [0, 1, 0, 14], [0, 0, 0, 1]], dtype=np.float32)   # quad stood up at z = 14
    sc.add_draw(sc.add_transform(wall), quad, albedo, INVALID)
    sc.add_draw(sc.add_transform(trs((-1.6, 0.9, 3.5), (0.9, 0.9, 0.9), 0.3)), sphere, albedo, mr)
    sc.add_draw(sc.add_transform(trs((1.2, 0.6, 2.2), (0.6, 0.6, 0.6), 1.1)), sphere, stone, mr)
    sc.add_draw(sc.add_transform(trs((0.3, 1.4, 6.0), (1.4, 1.4, 1.4), 2.0)), sphere, INVALID, INVALID)
    if cutout:
        fence_tex = _checker(128, 8, (40, 160, 60), (20, 110, 40), 4)
        yy, xx = np.mgrid[0:128, 0:128]
        fence_tex[((xx // 8) + (yy // 8)) % 3 == 0, 3] = 0  # holes of 8x8 texels
        fence = np.array([[3, 0, 0, 0.5], [0, 0, -1.2, 1.2], [0, 1, 0, 5], [0, 0, 0, 1]], dtype=np.float32)
        sc.add_draw(sc.add_transform(fence), quad, sc.add_texture(fence_tex), INVALID, flags=0xFF)
    return sc
