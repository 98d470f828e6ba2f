"""The post-process chain as a flat sequence of C-ABI calls on one set of images.

This is the thin Python driver used by the parity tests and `smoke()`: it allocates the
resource set of Gbuffer / GTAO / AdvancedSSR / TAA (scene_renderer.cpp:8-44, gtao.cpp:17-47,
advanced_ssr.cpp:62-92, taa.cpp:3-12) and issues the passes in the frame order of
main.cpp:345-391.  `backend` names a registered C-ABI provider: the package registers only
"product" (the HIP library, device memory); a checker exposing the same entry points on host memory
can be registered from outside with register_backend() (the test suite does).
The production host layer with the reference's pass structs lives in host/ (C++).
"""
import ctypes as C

import numpy as np

from . import abi
from .camera import FrameSetup
from .images import ImageBuf, depth_mip_count

PDF_LUT_SIZE = 1024

# name -> factory returning (ctypes library, entry-point prefix, runs on a device stream?)
_BACKENDS = {"product": lambda: (abi.product(), "vkr_", True)}


def register_backend(name, factory):
    """factory() -> (lib, prefix, on_device).  on_device False: images live in host memory and calls take no stream."""
    if name == "product":
        raise ValueError("the product backend cannot be replaced")
    _BACKENDS[name] = factory


class PostFxChain:
    def __init__(self, width, height, backend="product", device=None, setup=None, window=None, force_tiled=None):
        """window: None (single GPU) or (ox, oy, w, h) full-res window of the frame held by this instance."""
        self.W, self.H = width, height
        self.backend = backend
        if backend not in _BACKENDS:
            raise RuntimeError(f"backend {backend!r} is not registered (this package ships only 'product')")
        self.lib, self.prefix, on_device = _BACKENDS[backend]()
        if on_device:
            if device is None:
                device = "cuda"
            self.stream = self._stream_ptr(device)
        else:
            device = None
            self.stream = None
        self.device = device
        self.setup = setup or FrameSetup(width, height)
        if window is None:
            window = (0, 0, width, height)
        ox, oy, ww, wh = window
        assert ox % 2 == 0 and oy % 2 == 0 and ww % 2 == 0 and wh % 2 == 0
        self.window = window
        L = depth_mip_count(width, height)
        self.L = L
        full, org = (width, height), (ox, oy)
        half, horg = (width // 2, height // 2), (ox // 2, oy // 2)
        w2, h2 = ww // 2, wh // 2

        def img(fmt, w, h, mips=1, f=full, o=org, fill=0):
            return ImageBuf(fmt, w, h, mips, device=device, full=f, origin=o, fill=fill)

        # Gbuffer (scene_renderer.cpp:8-44).  Tiled: SSR marches `frame_hiz` (whole-frame image mips 1..L-1)
        # instead of the window-local chain, whose mips >= 2 are then unused.
        tiled = (ww, wh) != (width, height) if force_tiled is None else bool(force_tiled)
        self.tiled = tiled
        dm = depth_mip_count(ww, wh)  # Gbuffer(graph, w, h) sizes the chain from the extent it is given
        self.depth = img(abi.FMT_D24_UNORM_S8, ww, wh, dm)
        self.prev_depth = img(abi.FMT_D24_UNORM_S8, ww, wh, dm)
        self.normal = img(abi.FMT_RG16_UNORM, ww, wh)
        self.albedo = img(abi.FMT_RGBA8_SRGB, ww, wh)
        self.material = img(abi.FMT_RGBA8_SRGB, ww, wh)
        self.velocity = img(abi.FMT_RG16_SFLOAT, ww, wh)
        self.dn = img(abi.FMT_RG16_UNORM, w2, h2, f=half, o=horg)
        self.dv = img(abi.FMT_RG16_SFLOAT, w2, h2, f=half, o=horg)
        # GTAO (gtao.cpp:26-38)
        self.raw = img(abi.FMT_RGBA16_SFLOAT, w2, h2, f=half, o=horg)
        self.filtered = img(abi.FMT_R16_SFLOAT, w2, h2, f=half, o=horg)
        self.acc_ao = img(abi.FMT_RG16_SFLOAT, w2, h2, f=half, o=horg)
        self.acc_hist = img(abi.FMT_RG16_SFLOAT, w2, h2, f=half, o=horg)
        # AdvancedSSR (advanced_ssr.cpp:62-92)
        self.rays = img(abi.FMT_RGBA16_UNORM, w2, h2, f=half, o=horg)
        self.reflections = img(abi.FMT_RGBA8_UNORM, w2, h2, f=half, o=horg)
        self.blurred = img(abi.FMT_RGBA8_UNORM, w2, h2, f=half, o=horg)
        self.blurred_hist = img(abi.FMT_RGBA8_UNORM, w2, h2, f=half, o=horg)
        self.pdf = ImageBuf(abi.FMT_R32_SFLOAT, PDF_LUT_SIZE, PDF_LUT_SIZE, device=device)
        self.halton = self._make_halton()
        # TAA (taa.cpp:6-9)
        self.taa_hist = img(abi.FMT_RGBA16_SFLOAT, ww, wh)
        self.taa_target = img(abi.FMT_RGBA16_SFLOAT, ww, wh)
        self.frame_index = 0
        if tiled:
            # whole-frame copies for reads with unbounded reach, filled by tiling.TiledFrame
            self.frame_hiz = ImageBuf(abi.FMT_D24_UNORM_S8, width // 2, height // 2, L - 1, device=device)
            self.frame_normals = ImageBuf(abi.FMT_RG16_UNORM, width // 2, height // 2, device=device)
            self.frame_albedo = ImageBuf(abi.FMT_RGBA8_SRGB, width, height, device=device)

    # ---- plumbing -------------------------------------------------------------------
    @staticmethod
    def _stream_ptr(device):
        import torch

        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def _make_halton(self):
        """advanced_ssr.cpp:22-34,54-58: 128 x vec4 UBO, xy = Halton(2,3) of index i+1 (float-floor quirk)."""
        def elem(index, base):
            f, r, cur = np.float32(1.0), np.float32(0.0), index
            while True:
                f = np.float32(f / np.float32(base))
                r = np.float32(r + np.float32(f * np.float32(cur % base)))
                cur = int(np.floor(np.float32(np.float32(cur) / np.float32(base))))
                if cur <= 0:
                    break
            return r

        h = np.zeros((abi.HALTON_SEQ_SIZE, 4), dtype=np.float32)
        for i in range(abi.HALTON_SEQ_SIZE):
            h[i, 0], h[i, 1] = elem(i + 1, 2), elem(i + 1, 3)
        self.halton_host = h
        if self.device is None:
            return h
        # the HIP trace kernel wants cos/sin(2*PI*y) in zw (vkr_halton23_fill, include/vkr_postfx.h)
        filled = np.zeros((abi.HALTON_SEQ_SIZE, 4), dtype=np.float32)
        self.lib.vkr_halton23_fill(filled.ctypes.data_as(C.c_void_p), abi.HALTON_SEQ_SIZE)
        assert np.array_equal(filled[:, :2], h[:, :2])
        h = filled
        import torch

        return torch.from_numpy(h).to(self.device)

    def _halton_ptr(self):
        if self.device is None:
            return C.c_void_p(self.halton.ctypes.data)
        return C.c_void_p(self.halton.data_ptr())

    def call(self, name, *args):
        fn = getattr(self.lib, self.prefix + name)
        if self.stream is not None:
            rc = fn(*args, self.stream)
        else:
            rc = fn(*args)
        abi.check(rc, self.lib if self.backend == "product" else None)

    def sync(self):
        if self.device is not None:
            import torch

            torch.cuda.synchronize(self.device)

    # ---- G-buffer ---------------------------------------------------------------------
    def synth(self):
        s = self.setup
        p = s.synth()
        self.call("synth_gbuffer", C.byref(self.depth.desc(0, 1)), C.byref(self.normal.desc()), C.byref(self.albedo.desc()),
                  C.byref(self.material.desc()), C.byref(self.velocity.desc()), C.byref(p))
        pp = s.synth(prev=True, depth_only=True)
        self.call("synth_gbuffer", C.byref(self.prev_depth.desc(0, 1)), None, None, None, None, C.byref(pp))

    def raster(self, scene, target="current"):
        """SceneRenderer::draw_taa (scene_renderer.cpp:140-220) on `scene` (scene.Scene) instead of the synthetic
        generator: fills albedo / normal / material / velocity / depth mip 0 for the current camera.  target="prev":
        only the depth of the previous camera is kept (into prev_depth), like build_prev_hiz does for the generator."""
        key = "_scene_dev" if self.device is not None else "_scene_host"
        cached = getattr(scene, key, None)
        if cached is None:
            cached = scene.upload(self.device)
            setattr(scene, key, cached)
        rs, _ = cached
        c = abi.GbufConst()
        st = self.setup
        if target == "current":
            c.view_projection, c.prev_view_projection = abi.Mat4.from_np(st.mvp), abi.Mat4.from_np(st.prev_mvp)
        else:
            c.view_projection, c.prev_view_projection = abi.Mat4.from_np(st.prev_mvp), abi.Mat4.from_np(st.prev_mvp)
        c.fovy_aspect_znear_zfar = (C.c_float * 4)(*[float(v) for v in st.fazz])
        depth = self.depth if target == "current" else self.prev_depth
        if target == "current":
            outs = (self.albedo, self.normal, self.material, self.velocity)
        else:
            if not hasattr(self, "_raster_dummy"):
                self._raster_dummy = [ImageBuf(f, self.albedo.width, self.albedo.height, device=self.device, full=self.albedo.full, origin=self.albedo.origin)
                                      for f in (abi.FMT_RGBA8_SRGB, abi.FMT_RG16_UNORM, abi.FMT_RGBA8_SRGB, abi.FMT_RG16_SFLOAT)]
            outs = self._raster_dummy
        scratch_ptr, nbytes = C.c_void_p(0), 0
        if self.backend == "product":
            import torch

            W, H = self.albedo.full
            nbytes = int(self.lib.vkr_raster_scratch_bytes(W, H, sum(d["index_count"] // 3 for d in scene.draws)))
            if getattr(self, "_raster_scratch", None) is None or self._raster_scratch.numel() < nbytes:
                self._raster_scratch = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            scratch_ptr = C.c_void_p(self._raster_scratch.data_ptr())
        self.call("raster_gbuffer", C.byref(rs), C.byref(c), C.byref(outs[0].desc()), C.byref(outs[1].desc()), C.byref(outs[2].desc()),
                  C.byref(outs[3].desc()), C.byref(depth.desc(0, 1)), scratch_ptr, nbytes)

    def build_prev_hiz(self):
        """prev_depth's mips: what last frame's DownsamplePass left in the image that is now prev_depth (main.cpp:416)."""
        half = dict(device=self.device, full=self.dn.full, origin=self.dn.origin)       # the window's geometry (tiled chains)
        full = dict(device=self.device, full=self.normal.full, origin=self.normal.origin)
        scratch_n = ImageBuf(abi.FMT_RG16_UNORM, self.dn.width, self.dn.height, **half)
        scratch_v = ImageBuf(abi.FMT_RG16_SFLOAT, self.dn.width, self.dn.height, **half)
        n0 = ImageBuf(abi.FMT_RG16_UNORM, self.normal.width, self.normal.height, **full)
        v0 = ImageBuf(abi.FMT_RG16_SFLOAT, self.normal.width, self.normal.height, **full)
        self.call("downsample_gbuffer", C.byref(self.prev_depth.desc()), C.byref(n0.desc()), C.byref(v0.desc()),
                  C.byref(scratch_n.desc()), C.byref(scratch_v.desc()))
        self.call("depth_mips", C.byref(self.prev_depth.desc()), 1)
        self.sync()

    def init_histories(self):
        """TAA history = current colour, AO history (1, 1/255), SSR history 0."""
        col = self.albedo.decode()[..., :3]
        th = np.zeros((self.albedo.height, self.albedo.width, 4), dtype=np.float16)
        th[..., :3] = col.astype(np.float16)
        host = ImageBuf(abi.FMT_RGBA16_SFLOAT, self.taa_hist.width, self.taa_hist.height)
        host.set_raw(th)
        self.taa_hist.upload(host.host)
        ah = np.zeros((self.acc_hist.height, self.acc_hist.width, 2), dtype=np.float16)
        ah[..., 0] = 1.0
        ah[..., 1] = np.float16(1.0 / 255.0)
        host = ImageBuf(abi.FMT_RG16_SFLOAT, self.acc_hist.width, self.acc_hist.height)
        host.set_raw(ah)
        self.acc_hist.upload(host.host)

    # ---- passes, in frame order (main.cpp:347-391) ----------------------------------------
    def downsample(self):
        self.call("downsample_gbuffer", C.byref(self.depth.desc()), C.byref(self.normal.desc()), C.byref(self.velocity.desc()),
                  C.byref(self.dn.desc()), C.byref(self.dv.desc()))
        self.call("depth_mips", C.byref(self.depth.desc()), 1)

    def hiz_tail(self, gathered_mips):
        """tiled: view mips 0..gathered-1 of frame_hiz arrived by all-gather; rebuild the rest locally"""
        self.call("depth_mips", C.byref(self.frame_hiz.desc()), gathered_mips - 1)

    def preintegrate_pdf(self):
        self.call("pdf_preintegrate", C.byref(self.pdf.desc()))

    def ssr_trace(self, frame_random=None, max_roughness=1.0, split=None):
        """split: None = one launch (vkr_sssr_trace); 0..4 = vkr_sssr_trace_split with that many compacted rounds in the
        head launch (product backend only: the images are the same bit for bit, so the oracle has no counterpart)."""
        tp = self.setup.trace_params(frame_random)
        push = abi.TracePush(max_roughness)
        # advanced_ssr.cpp:186: depth view = mips 1..L-1 (tiled: the gathered whole-frame pyramid)
        hiz = self.frame_hiz.desc() if self.tiled else self.depth.desc(1, self.depth.mips - 1)
        dn = self.frame_normals.desc() if self.tiled else self.dn.desc()
        if split is not None and self.backend == "product":
            import torch

            need = int(self.lib.vkr_sssr_trace_workspace_bytes(self.rays.width, self.rays.height))
            if getattr(self, "_trace_workspace", None) is None or self._trace_workspace.numel() < need:
                self._trace_workspace = torch.zeros(need, dtype=torch.uint8, device=self.device)
            self.call("sssr_trace_split", C.byref(hiz), C.byref(dn), C.byref(self.material.desc()), C.byref(tp), self._halton_ptr(),
                      C.byref(self.rays.desc()), C.byref(self.raw.desc()), C.byref(self.pdf.desc()), C.byref(push),
                      self._trace_workspace.data_ptr(), need, int(split))
            return
        self.call("sssr_trace", C.byref(hiz), C.byref(dn),
                  C.byref(self.material.desc()), C.byref(tp), self._halton_ptr(), C.byref(self.rays.desc()),
                  C.byref(self.raw.desc()), C.byref(self.pdf.desc()), C.byref(push))

    # ---- multi-GPU: hit colours / hit normals by request / reply (include/vkr_postfx.h; no reference counterpart) ----
    def _pending_images(self):
        if not hasattr(self, "pend_mask"):
            r = self.rays
            self.pend_mask = ImageBuf(abi.FMT_R8_UNORM, r.width, r.height, device=self.device, full=r.full, origin=r.origin)
            self.pend_data = ImageBuf(abi.FMT_RGBA32_SFLOAT, 2 * r.width, r.height, device=self.device)
        return self.pend_mask, self.pend_data

    def ssr_trace_windowed(self, frame_random=None, max_roughness=1.0):
        """vkr_sssr_trace_windowed: frame_normals holds only this window's rows; rays whose hit-normal footprint leaves them
        stay pending (pend_mask / pend_data) for ssr_validate()."""
        assert self.tiled
        mask, data = self._pending_images()
        tp = self.setup.trace_params(frame_random)
        oy, h2 = self.dn.origin[1], self.dn.height
        push = abi.TraceWindowPush(max_roughness, oy, oy + h2)
        self.call("sssr_trace_windowed", C.byref(self.frame_hiz.desc()), C.byref(self.frame_normals.desc()), C.byref(self.material.desc()),
                  C.byref(tp), self._halton_ptr(), C.byref(self.rays.desc()), C.byref(self.raw.desc()), C.byref(self.pdf.desc()),
                  C.byref(mask.desc()), C.byref(data.desc()), C.byref(push))

    def _trace_ws(self):
        import torch

        need = int(self.lib.vkr_sssr_trace_workspace_bytes(self.rays.width, self.rays.height))
        if getattr(self, "_trace_workspace", None) is None or self._trace_workspace.numel() < need:
            self._trace_workspace = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return self._trace_workspace.data_ptr(), need

    def _windowed_args(self, frame_random, max_roughness):
        mask, data = self._pending_images()
        tp = self.setup.trace_params(frame_random)
        oy, h2 = self.dn.origin[1], self.dn.height
        push = abi.TraceWindowPush(max_roughness, oy, oy + h2)
        self._wargs = (self.frame_hiz.desc(), self.frame_normals.desc(), self.material.desc(), tp, self.rays.desc(), self.raw.desc(), self.pdf.desc(),
                       mask.desc(), data.desc(), push)  # kept alive across the call
        return self._wargs

    def ssr_trace_windowed_head(self, local_levels, frame_random=None, max_roughness=1.0, park_after=2):
        """vkr_sssr_trace_windowed_head: marches on the window image's own pyramid levels 1..local_levels (product backend only)"""
        assert self.tiled and self.backend == "product"
        hiz, nrm, mat, tp, rays, raw, pdf, mask, data, push = self._windowed_args(frame_random, max_roughness)
        local = self.depth.desc(1, local_levels)
        ws, need = self._trace_ws()
        self.call("sssr_trace_windowed_head", C.byref(local), C.byref(hiz), C.byref(nrm), C.byref(mat), C.byref(tp), self._halton_ptr(),
                  C.byref(rays), C.byref(raw), C.byref(pdf), C.byref(mask), C.byref(data), C.byref(push), ws, need, int(park_after))

    def ssr_trace_windowed_resume(self, frame_random=None, max_roughness=1.0):
        assert self.tiled and self.backend == "product"
        hiz, nrm, mat, tp, rays, raw, pdf, mask, data, push = self._windowed_args(frame_random, max_roughness)
        ws, need = self._trace_ws()
        self.call("sssr_trace_windowed_resume", C.byref(hiz), C.byref(nrm), C.byref(mat), C.byref(tp), self._halton_ptr(),
                  C.byref(rays), C.byref(raw), C.byref(pdf), C.byref(mask), C.byref(data), C.byref(push), ws, need)

    def ssr_validate(self):
        mask, data = self._pending_images()
        tp = self.setup.trace_params()
        self.call("sssr_validate", C.byref(self.rays.desc()), C.byref(mask.desc()), C.byref(data.desc()), C.byref(self.frame_normals.desc()), C.byref(tp))

    def _u32_buffer(self, n):
        if self.device is None:
            return np.zeros(max(n, 1), dtype=np.uint32)
        import torch

        return torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)

    def _hit_sources(self, normals):
        self._hit_descs = [self.rays.desc()]  # keep the descriptors alive while the struct points at them
        src = abi.HitSources()
        src.rays = C.pointer(self._hit_descs[0])
        src.albedo_width, src.albedo_height = self.W, self.H
        src.window_row0, src.window_row1 = self.window[1], self.window[1] + self.window[3]
        if normals:
            mask, data = self._pending_images()
            self._hit_descs += [mask.desc(), data.desc()]
            src.pending_mask, src.pending_data = C.pointer(self._hit_descs[1]), C.pointer(self._hit_descs[2])
            src.normal_width, src.normal_height = self.W // 2, self.H // 2
            src.normal_row0, src.normal_row1 = self.dn.origin[1], self.dn.origin[1] + self.dn.height
        return src

    def hit_count(self, row_bounds, normals=True):
        """[requests this window has for owner o] (vkr_hit_requests pass 1)"""
        world = len(row_bounds) - 1
        counts = self._u32_buffer(world)
        self._hit_workspace = self._u32_buffer(abi.HIT_WORKSPACE_WORDS)  # pass 1 leaves its per-block counts here for pass 2
        b = (C.c_uint32 * (world + 1))(*row_bounds)
        self.call("hit_requests", C.byref(self._hit_sources(normals)), b, world, self._buf_ptr(counts), self._buf_ptr(self._hit_workspace), None, None)
        return [int(v) for v in self.buffer_to_host(counts)[:world]]

    def hit_write(self, row_bounds, counts, normals=True):
        """The requests, owner o's at [segments[o], segments[o + 1]) (vkr_hit_requests pass 2) -> (uint32 buffer, segments)"""
        world = len(row_bounds) - 1
        segments = [0]
        for c in counts:
            segments.append(segments[-1] + c)
        out = self._u32_buffer(segments[-1])
        b = (C.c_uint32 * (world + 1))(*row_bounds)
        seg = (C.c_uint32 * world)(*segments[:world])
        self.call("hit_requests", C.byref(self._hit_sources(normals)), b, world, None, self._buf_ptr(self._hit_workspace), seg, self._buf_ptr(out))
        return out, segments

    def hit_write_bounded(self, row_bounds, capacities, normals=True):
        """vkr_hit_requests_bounded after hit_count(): segments of fixed room (capacities per owner), unused slots VKR_HIT_NO_REQUEST
        -> (uint32 buffer, segments, dropped flag).  Product backend only."""
        import torch

        assert self.backend == "product"
        world = len(row_bounds) - 1
        segments = [0]
        for c in capacities:
            segments.append(segments[-1] + c)
        out = torch.full((max(segments[-1], 1),), -1, dtype=torch.int32, device=self.device)  # 0xFFFFFFFF: no request
        dropped = self._u32_buffer(1)
        b = (C.c_uint32 * (world + 1))(*row_bounds)
        seg = (C.c_uint32 * (world + 1))(*segments)
        cap = (C.c_uint32 * world)(*capacities)
        self.call("hit_requests_bounded", C.byref(self._hit_sources(normals)), b, world, self._buf_ptr(self._hit_workspace), seg, cap, self._buf_ptr(out),
                  self._buf_ptr(dropped))
        return out, segments, int(self.buffer_to_host(dropped)[0])

    def hit_reply(self, requests, count, normals=True):
        """16 bytes per request from this window's albedo / downsampled normals -> (uint32 buffer [4 * count], errors)"""
        replies, errors = self._u32_buffer(4 * count), self._u32_buffer(1)
        dn = self.dn.desc()
        self.call("hit_reply", C.byref(self.albedo.desc()), C.byref(dn) if normals else None, self._buf_ptr(requests), count,
                  self._buf_ptr(replies), self._buf_ptr(errors))
        return replies, int(self.buffer_to_host(errors)[0])

    def hit_scatter(self, requests, replies, count, normals=True):
        fn = self.frame_normals.desc()
        self.call("hit_scatter", C.byref(self.frame_albedo.desc()), C.byref(fn) if normals else None, self._buf_ptr(requests),
                  self._buf_ptr(replies), count)

    def ssr_filter(self, render_flags=7):
        tp = self.setup.trace_params()
        push = abi.FilterPush(render_flags)
        nm = min(10, self.depth.mips)  # advanced_ssr.cpp:342: mips 0..9
        albedo = self.frame_albedo if self.tiled else self.albedo
        self.call("sssr_filter", C.byref(self.rays.desc()), C.byref(self.depth.desc(0, nm)), C.byref(albedo.desc()),
                  C.byref(self.normal.desc()), C.byref(self.material.desc()), C.byref(self.reflections.desc()),
                  C.byref(tp), C.byref(push))

    def ssr_blur(self, max_roughness=1.0, accumulate=1, disable_blur=0):
        rp = self.setup.reproject_params()
        push = abi.BlurPush(max_roughness, accumulate, disable_blur)
        nm = min(10, self.depth.mips)
        self.call("sssr_blur", C.byref(self.depth.desc(0, nm)), C.byref(self.normal.desc()), C.byref(self.reflections.desc()),
                  C.byref(self.material.desc()), C.byref(self.blurred_hist.desc()), C.byref(self.dv.desc()),
                  C.byref(self.prev_depth.desc(0, nm)), C.byref(self.blurred.desc()), C.byref(rp), C.byref(push))

    def gtao_main(self, **push_kw):
        gp = self.setup.gtao_params()
        push = self.setup.gtao_push(**push_kw)
        # gtao.cpp:119: depth image-mip 1 bound as a 1-mip view
        self.call("gtao_main", C.byref(self.depth.desc(1, 1)), C.byref(gp), C.byref(self.normal.desc()),
                  C.byref(self.material.desc()), C.byref(self.pdf.desc()), C.byref(self.raw.desc()), C.byref(push))

    def gtao_filter(self):
        push = self.setup.gtao_filter_push()
        self.call("gtao_filter", C.byref(self.depth.desc(1, 1)), C.byref(self.raw.desc()), C.byref(self.filtered.desc()),
                  C.byref(push))

    def gtao_accumulate(self, clear_history=0):
        ap = self.setup.gtao_accum_params()
        push = abi.GtaoAccumPush(clear_history)
        self.call("gtao_accumulate", C.byref(self.depth.desc(1, 1)), C.byref(self.prev_depth.desc(1, 1)),
                  C.byref(self.filtered.desc()), C.byref(self.acc_ao.desc()), C.byref(self.dv.desc()),
                  C.byref(self.acc_hist.desc()), C.byref(ap), C.byref(push))

    def taa(self, color=None):
        rp = self.setup.reproject_params()
        color = color or self.albedo  # colour input until the deferred composite exists
        self.call("taa_resolve", C.byref(self.taa_hist.desc()), C.byref(self.prev_depth.desc()), C.byref(self.depth.desc()),
                  C.byref(self.velocity.desc()), C.byref(color.desc()), C.byref(self.taa_target.desc()), C.byref(rp))

    def ssr_simple(self, color=None):
        """src/ssr.cpp add_ssr_pass: full-res mirror SSR into an RGBA8_UNORM target (create_ssr_tex)."""
        if not hasattr(self, "ssr_out"):
            self.ssr_out = ImageBuf(abi.FMT_RGBA8_UNORM, self.albedo.width, self.albedo.height, device=self.device,
                                    full=self.albedo.full, origin=self.albedo.origin)
        p = abi.SsrParams()
        p.normal_mat = abi.Mat4.from_np(self.setup.normal_mat)
        p.fovy, p.aspect, p.znear, p.zfar = [float(v) for v in self.setup.fazz]
        color = color or self.albedo
        self.call("ssr", C.byref(self.normal.desc()), C.byref(self.depth.desc()), C.byref(color.desc()), C.byref(p),
                  C.byref(self.material.desc()), C.byref(self.ssr_out.desc()))

    def preintegrate_brdf(self):
        if not hasattr(self, "brdf"):
            self.brdf = ImageBuf(abi.FMT_RG16_SFLOAT, PDF_LUT_SIZE, PDF_LUT_SIZE, device=self.device)
        self.call("brdf_preintegrate", self._halton_ptr(), C.byref(self.brdf.desc()))

    def shading(self, min_roughness=0.0, max_roughness=1.0, show_ao=0):
        """src/defered_shading.cpp DeferedShadingPass::draw -> color_out (RGBA8_SRGB); SURVEY.md 8(f) #1."""
        if not hasattr(self, "color_out"):
            self.color_out = ImageBuf(abi.FMT_RGBA8_SRGB, self.albedo.width, self.albedo.height, device=self.device,
                                      full=self.albedo.full, origin=self.albedo.origin)
        p = abi.ShadingParams()
        p.inverse_camera = abi.Mat4.from_np(self.setup.inv_view)
        p.camera = abi.Mat4.from_np(self.setup.view)
        p.shadow_mvp = abi.Mat4.from_np(np.eye(4))
        p.fovy, p.aspect, p.znear, p.zfar = [float(v) for v in self.setup.fazz]
        push = abi.ShadingPush((C.c_float * 2)(min_roughness, max_roughness), show_ao)
        self.call("defered_shading", C.byref(self.albedo.desc()), C.byref(self.normal.desc()), C.byref(self.material.desc()),
                  C.byref(self.depth.desc()), C.byref(p), C.byref(self.acc_ao.desc()), C.byref(self.brdf.desc()),
                  C.byref(self.blurred.desc()), C.byref(self.color_out.desc()), C.byref(push))

    # ---- tile-classified trace (advanced_ssr.cpp:216-302,440-495; commented out of AdvancedSSR::run) ----
    def _i32_buffer(self, n):
        if self.device is None:
            return np.zeros(n, dtype=np.int32)
        import torch

        return torch.zeros(n, dtype=torch.int32, device=self.device)

    @staticmethod
    def _buf_ptr(b):
        return C.c_void_p(b.ctypes.data if isinstance(b, np.ndarray) else b.data_ptr())

    def buffer_to_host(self, b):
        if isinstance(b, np.ndarray):
            return b.copy()
        self.sync()
        return b.cpu().numpy()

    def ssr_classify(self, max_roughness=1.0, glossy_value=0.5):
        """SSSR_Clear + SSSR_Classification: fills reflective/glossy tile lists and their indirect arguments."""
        w2, h2 = self.rays.full
        if not hasattr(self, "reflective_tiles"):
            # advanced_ssr.cpp:81-83: sizeof(int) * (w*h/64) with the full-res extent
            cap = max(1, (w2 * 2) * (h2 * 2) // 64)
            self.tile_capacity = cap
            self.reflective_tiles, self.glossy_tiles = self._i32_buffer(cap), self._i32_buffer(cap)
            self.reflective_args, self.glossy_args = self._i32_buffer(3), self._i32_buffer(3)
        self.call("sssr_clear_indirect", self._buf_ptr(self.reflective_args), self._buf_ptr(self.glossy_args))
        push = abi.ClassificationPush(w2, h2, max_roughness, glossy_value)
        self.call("sssr_classification", C.byref(self.material.desc()), self._buf_ptr(self.reflective_tiles), self._buf_ptr(self.glossy_tiles),
                  self._buf_ptr(self.reflective_args), self._buf_ptr(self.glossy_args), C.byref(push))

    def ssr_trace_indirect(self, frame_random=None, max_roughness=1.0):
        """run_trace_indirect_pass: mirror tiles then glossy tiles into `rays`."""
        tp = self.setup.trace_params(frame_random)
        dview = self.frame_hiz.desc() if self.tiled else self.depth.desc(1, self.depth.mips - 1)
        normal = self.frame_normals if self.tiled else self.dn
        for kind, tiles, args in ((0, self.reflective_tiles, self.reflective_args), (1, self.glossy_tiles, self.glossy_args)):
            push = abi.TraceIndirectPush(kind, max_roughness)
            self.call("sssr_trace_indirect", C.byref(dview), C.byref(normal.desc()), C.byref(self.material.desc()), C.byref(tp),
                      self._halton_ptr(), C.byref(self.rays.desc()), self._buf_ptr(tiles), self._buf_ptr(args), self.tile_capacity,
                      C.byref(push))

    # ---- passes the reference ships but never records ------------------
    def _half_img(self, fmt, name):
        if not hasattr(self, name):
            setattr(self, name, ImageBuf(fmt, self.raw.width, self.raw.height, device=self.device, full=self.raw.full, origin=self.raw.origin))
        return getattr(self, name)

    def gtao_main_graphics(self, angle_offset=60.0 / 360.0):
        """gtao.cpp:349-413 add_main_pass_graphics -> raw (program "gtao_main")."""
        push = abi.GtaoGfxPush(angle_offset)
        self.call("gtao_main_graphics", C.byref(self.depth.desc(1, 1)), C.byref(self.setup.gtao_params()), C.byref(self.normal.desc()),
                  C.byref(self.raw.desc()), C.byref(push))

    def gtao_reproject(self):
        """gtao.cpp:241-284 add_reprojection_pass: filtered + prev_frame -> output (all R16F)."""
        prev, out = self._half_img(abi.FMT_R16_SFLOAT, "ao_prev_frame"), self._half_img(abi.FMT_R16_SFLOAT, "ao_output")
        p = abi.GtaoReprojection()
        p.camera_to_prev_frame = abi.Mat4.from_np(np.eye(4))
        p.fovy, p.aspect, p.znear, p.zfar = [float(v) for v in self.setup.fazz]
        self.call("gtao_reproject", C.byref(p), C.byref(self.depth.desc(1, 1)), C.byref(self.prev_depth.desc(1, 1)),
                  C.byref(self.filtered.desc()), C.byref(prev.desc()), C.byref(out.desc()))

    def _layer_descs(self, pattern_n):
        step = 1 << pattern_n
        if not hasattr(self, "deint_layers"):
            self.deint_layers = [ImageBuf(abi.FMT_R32_SFLOAT, self.raw.width // step, self.raw.height // step, device=self.device)
                                 for _ in range(step * step)]
        arr = (abi.VkrImg * len(self.deint_layers))()
        for i, l in enumerate(self.deint_layers):
            arr[i] = l.desc()
        return arr

    def deinterleave_depth(self, pattern_n=2):
        """gtao.cpp:445-470 deinterleave_depth (program "deinterleave_depth")."""
        arr = self._layer_descs(pattern_n)
        self.call("deinterleave_depth", C.byref(self.depth.desc(1, 1)), arr, len(arr), C.byref(abi.DeinterleavePush(pattern_n)))

    def gtao_main_deinterleaved(self, layer=0, pattern_n=2, angle_offset=60.0 / 360.0):
        """gtao.cpp:472-526 add_main_pass_deinterleaved, one dispatch (program "main_deinterleaved")."""
        arr = self._layer_descs(pattern_n)
        push = abi.GtaoDeinterleavedPush(pattern_n, layer, angle_offset)
        self.call("gtao_main_deinterleaved", arr, len(arr), C.byref(self.setup.gtao_params()), C.byref(self.normal.desc()),
                  C.byref(self.raw.desc()), C.byref(push))

    def _full_img(self, name):
        if not hasattr(self, name):
            setattr(self, name, ImageBuf(abi.FMT_RGBA16_SFLOAT, self.albedo.width, self.albedo.height, device=self.device))
        return getattr(self, name)

    def screen_trace(self, angle_offset=60.0 / 360.0, random_offset=0.25, color=None):
        """screen_trace.cpp:23-95 ScreenSpaceTrace::add_main_pass -> st_raw (full-res RGBA16F)."""
        out = self._full_img("st_raw")
        p = abi.ScreenTraceParams()
        p.normal_mat = abi.Mat4.from_np(self.setup.normal_mat)
        p.random_offset, p.angle_offset = random_offset, angle_offset
        p.fovy, p.aspect, p.znear, p.zfar = [float(v) for v in self.setup.fazz]
        color = color or self.albedo
        self.call("screen_trace_main", C.byref(self.depth.desc(0, 1)), C.byref(self.normal.desc()), C.byref(color.desc()),
                  C.byref(self.material.desc()), C.byref(out.desc()), C.byref(p))

    def screen_trace_filter(self):
        """screen_trace.cpp:97-140: st_raw -> st_filtered."""
        znear, zfar = float(self.setup.fazz[2]), float(self.setup.fazz[3])
        self.call("screen_trace_filter", C.byref(self._full_img("st_raw").desc()), C.byref(self.depth.desc(0, 1)),
                  C.byref(self._full_img("st_filtered").desc()), C.byref(abi.ScreenTraceFilterPush(znear, zfar)))

    def screen_trace_accumulate(self):
        """screen_trace.cpp:142-181: st_filtered -> st_accumulated (in place)."""
        push = abi.ScreenTraceAccumPush(*[float(v) for v in self.setup.fazz])
        self.call("screen_trace_accumulate", C.byref(self.depth.desc(0, 1)), C.byref(self.prev_depth.desc(0, 1)),
                  C.byref(self._full_img("st_filtered").desc()), C.byref(self._full_img("st_accumulated").desc()), C.byref(push))

    def frame(self):
        """One steady-state frame of the chain: D1 D2 S1 S2 S3 G1 G2 G3 T (main.cpp:347-391)."""
        self.downsample()
        self.ssr_trace(frame_random=self.frame_index % 16)  # advanced_ssr.cpp:168-171
        self.ssr_filter()
        self.ssr_blur()
        self.gtao_main()
        self.gtao_filter()
        self.gtao_accumulate()
        self.taa()
        self.frame_index += 1

    def swap_histories(self):
        """main.cpp:416-420 minus depth<->prev_depth (the benchmark G-buffer is static)."""
        self.taa_hist, self.taa_target = self.taa_target, self.taa_hist
        self.blurred, self.blurred_hist = self.blurred_hist, self.blurred
        self.acc_ao, self.acc_hist = self.acc_hist, self.acc_ao

    OUTPUTS = ("dn", "dv", "depth", "rays", "raw", "reflections", "blurred", "filtered", "acc_ao", "taa_target")
