"""ctypes binding of the C++ host mirror (host/libvkr_host.so): the reference's pass structs
(Gbuffer, DownsamplePass, AdvancedSSR, GTAO, TAA) on the rendergraph mirror, driven one stage at
a time.  Device memory for every graph image comes from torch through the allocator hook, so
the launcher can hand the same memory to torch.distributed (RCCL) without copies."""
import ctypes as C
import os

import numpy as np

from . import abi

STAGE_LUT, STAGE_GBUFFER, STAGE_PREV_DEPTH, STAGE_DOWNSAMPLE = 1, 2, 4, 8
STAGE_HIZ_TAIL, STAGE_SSR, STAGE_GTAO, STAGE_TAA = 16, 32, 64, 128
STAGE_SHADING, STAGE_BRDF_LUT, STAGE_GTAO_MAIN_ONLY = 256, 512, 1024
STAGE_GTAO_GRAPHICS, STAGE_GTAO_DEINTERLEAVED, STAGE_SCREEN_TRACE = 2048, 4096, 8192
STAGE_SSR_CLASSIFIED, STAGE_SSR_TRACE, STAGE_SSR_RESOLVE = 16384, 32768, 65536
STAGE_RASTER = 131072
STAGE_CHAIN = STAGE_DOWNSAMPLE | STAGE_SSR | STAGE_GTAO | STAGE_TAA


class HostConfig(C.Structure):
    _fields_ = [("full_width", C.c_uint32), ("full_height", C.c_uint32), ("origin_x", C.c_int32), ("origin_y", C.c_int32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("tiled", C.c_uint32), ("stream", C.c_void_p)]


class TiledConfig(C.Structure):
    _fields_ = [("full_width", C.c_uint32), ("full_height", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("halo", C.c_uint32), ("gathered_mips", C.c_uint32), ("force_tiled", C.c_uint32), ("albedo_by_gather", C.c_uint32),
                ("stream", C.c_void_p), ("comm", C.c_void_p), ("row_bounds", C.POINTER(C.c_uint32))]


class GatherPart(C.Structure):
    _fields_ = [("send", C.c_void_p), ("recv", C.c_void_p), ("bytes", C.c_uint64)]


class HaloPeer(C.Structure):
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_uint32), ("send", C.c_void_p), ("send_bytes", C.c_uint64),
                ("recv", C.c_void_p), ("recv_bytes", C.c_uint64)]


class HostCamera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("prev_view", C.c_float * 16), ("projection", C.c_float * 16),
                ("fovy", C.c_float), ("aspect", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class SceneDraw(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("vertex_offset", C.c_uint32), ("index_offset", C.c_uint32), ("index_count", C.c_uint32),
                ("albedo_tex_index", C.c_uint32), ("metalic_roughness_index", C.c_uint32), ("clip_alpha", C.c_uint32)]


class SceneTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mip_levels", C.c_uint32), ("reserved", C.c_uint32),
                ("levels", C.c_void_p * 16)]


_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_uint64, C.c_void_p)
_FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_lib = None


def lib():
    global _lib
    if _lib is None:
        l = abi._load(abi.HOST_LIB, "host library libvkr_host.so")
        l.vkrh_create.argtypes = [C.POINTER(HostConfig)]
        l.vkrh_create.restype = C.c_void_p
        l.vkrh_destroy.argtypes = [C.c_void_p]
        l.vkrh_last_error.restype = C.c_char_p
        l.vkrh_set_camera.argtypes = [C.c_void_p, C.POINTER(HostCamera)]
        l.vkrh_pin_randoms.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32]
        l.vkrh_has_program.argtypes = [C.c_char_p]
        l.vkrh_set_gtao_mode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        l.vkrh_set_synth_flags.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_set_gathered_mips.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_run.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_end_frame.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_image.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(abi.VkrImg)]
        l.vkrh_image_layer.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(abi.VkrImg)]
        l.vkrh_pin_screen_trace.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint32]
        l.vkrh_capture.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p]
        l.vkrh_read_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        l.vkrh_load_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(SceneDraw), C.c_uint32,
                                      C.POINTER(SceneTexture), C.c_uint32]
        l.vkrh_last_tasks.argtypes = [C.c_void_p]
        l.vkrh_last_tasks.restype = C.c_char_p
        l.vkrh_last_lanes.argtypes = [C.c_void_p]
        l.vkrh_last_lanes.restype = C.c_char_p
        l.vkrh_set_async.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_enable_task_timing.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_enable_task_timing_only.argtypes = [C.c_void_p, C.c_char_p]
        l.vkrh_collect_task_times.argtypes = [C.c_void_p]
        l.vkrh_collect_task_times.restype = C.c_char_p
        l.vkrh_set_allocator.argtypes = [_ALLOC, _FREE, C.c_void_p]
        l.vkrh_tiled_create.argtypes = [C.POINTER(TiledConfig)]
        l.vkrh_tiled_create.restype = C.c_void_p
        l.vkrh_tiled_destroy.argtypes = [C.c_void_p]
        l.vkrh_tiled_frame.argtypes = [C.c_void_p]
        l.vkrh_tiled_frame.restype = C.c_void_p
        l.vkrh_tiled_step.argtypes = [C.c_void_p]
        l.vkrh_tiled_flush.argtypes = [C.c_void_p]
        l.vkrh_tiled_phase.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_tiled_gather_parts.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(GatherPart), C.c_uint32, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_halo_peers.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(HaloPeer), C.c_uint32, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_hit_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_hit_requests.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(HaloPeer), C.c_uint32, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_hit_replies.argtypes = [C.c_void_p, C.POINTER(HaloPeer), C.c_uint32, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_hit_finish.argtypes = [C.c_void_p]
        l.vkrh_tiled_hit_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        l.vkrh_tiled_hit_rounds.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        l.vkrh_tiled_local_first.argtypes = [C.c_void_p]
        l.vkrh_hit_capacities.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_emulate_wire.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_hit_errors.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        l.vkrh_tiled_time_waits.argtypes = [C.c_void_p, C.c_uint32]
        l.vkrh_tiled_wait_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.vkrh_balance_rows.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        l.vkrh_balance_rows.restype = C.c_int
        _lib = l
    return _lib


def hit_capacities(counts, world, percent=125):
    """frame.hpp vkrh_hit_capacities: the segment room of the next frame's hit round from this frame's world x world counts"""
    flat = (C.c_uint32 * len(counts))(*counts)
    out = (C.c_uint32 * len(counts))()
    if lib().vkrh_hit_capacities(flat, world, percent, out) != 0:
        raise RuntimeError("host error: " + lib().vkrh_last_error().decode())
    return list(out)


def balance_rows(ms, bounds, align=16, min_rows=64):
    """frame.hpp vkrh_balance_rows: new strip bounds from the compute time each rank measured with `bounds`."""
    world = len(ms)
    assert len(bounds) == world + 1
    out = (C.c_uint32 * (world + 1))()
    rc = lib().vkrh_balance_rows((C.c_float * world)(*[float(v) for v in ms]), (C.c_uint32 * (world + 1))(*bounds), world, align, min_rows, out)
    if rc != 0:
        raise RuntimeError("vkrh_balance_rows: " + lib().vkrh_last_error().decode())
    return list(out)


class TorchAllocator:
    """Backs graph images with torch uint8 tensors (keeps them alive until the graph frees them)."""

    def __init__(self, device):
        import torch

        self.torch = torch
        self.device = device
        self.live = {}
        self._alloc = _ALLOC(self.alloc)
        self._free = _FREE(self.free)

    def alloc(self, nbytes, _user):
        t = self.torch.zeros((int(nbytes),), dtype=self.torch.uint8, device=self.device)
        self.live[t.data_ptr()] = t
        return t.data_ptr()

    def free(self, ptr, _user):
        self.live.pop(ptr, None)

    def install(self):
        lib().vkrh_set_allocator(self._alloc, self._free, None)

    def tensor_at(self, ptr):
        """(tensor, byte offset) of the allocation containing `ptr`."""
        for base, t in self.live.items():
            if base <= ptr < base + t.numel():
                return t, ptr - base
        raise KeyError(hex(ptr))


def _mat16(m):
    a = (C.c_float * 16)()
    flat = np.asarray(m, dtype=np.float32).T.reshape(-1)
    for i in range(16):
        a[i] = float(flat[i])
    return a


class HostFrame:
    def __init__(self, setup, device="cuda", window=None, tiled=False, stream=None, native_tiled=None):
        """native_tiled: None, or dict(rank, world, halo, gathered_mips, force_tiled, comm) — the frame then lives inside
        the C++ tiled frame (frame.hpp vkrh_tiled_*: strips, exchanges issued from C++ on their own stream); `window` is
        derived there."""
        import torch

        self.setup = setup
        W, H = setup.width, setup.height
        ox, oy, ww, wh = window or (0, 0, W, H)
        self.allocator = TorchAllocator(device)
        self.allocator.install()
        if stream is None:
            stream = torch.cuda.current_stream(device).cuda_stream
        self.tiled_handle = None
        self.albedo_by_gather, self.gather_mode = True, 1  # only the C++ tiled frame has the request / reply exchanges
        if native_tiled is not None:
            nt = native_tiled
            comm = nt.get("comm")
            bounds = nt.get("row_bounds")  # world + 1 strip boundaries (rows), or None for equal strips
            arr = (C.c_uint32 * len(bounds))(*bounds) if bounds is not None else None
            # 0 (default): hit colours and hit normals by request / reply; 1: albedo and normals all-gathered (round 2);
            # 2: albedo by request, normals gathered
            mode = nt.get("gather_mode")
            if mode is None:
                mode = int(os.environ.get("VKR_TILED_GATHER_MODE", "1" if os.environ.get("VKR_TILED_ALBEDO_GATHER") == "1" else "0"))
            self.gather_mode = int(mode)
            self.albedo_by_gather = self.gather_mode == 1
            tc = TiledConfig(W, H, nt["rank"], nt["world"], nt["halo"], nt["gathered_mips"], 1 if nt.get("force_tiled") else 0, self.gather_mode,
                             C.c_void_p(stream), C.c_void_p(comm.handle if comm is not None else None),
                             C.cast(arr, C.POINTER(C.c_uint32)) if arr is not None else None)
            self.comm = comm  # keep the communicator alive as long as the frame
            self.tiled_handle = lib().vkrh_tiled_create(C.byref(tc))
            if not self.tiled_handle:
                raise RuntimeError("vkrh_tiled_create failed: " + lib().vkrh_last_error().decode())
            self.h = lib().vkrh_tiled_frame(self.tiled_handle)
        else:
            cfg = HostConfig(W, H, ox, oy, ww, wh, 1 if tiled else 0, C.c_void_p(stream))
            self.h = lib().vkrh_create(C.byref(cfg))
            if not self.h:
                raise RuntimeError("vkrh_create failed: " + lib().vkrh_last_error().decode())
        cam = HostCamera()
        cam.view, cam.prev_view, cam.projection = _mat16(setup.view), _mat16(setup.prev_view), _mat16(setup.proj)
        cam.fovy, cam.aspect, cam.znear, cam.zfar = [float(v) for v in setup.fazz]
        self._check(lib().vkrh_set_camera(self.h, C.byref(cam)))
        self.pin_randoms()
        self._check(lib().vkrh_set_gtao_mode(self.h, setup.use_mis, 0))
        self._check(lib().vkrh_set_synth_flags(self.h, getattr(setup, "synth_flags", 0)))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("host frame error: " + lib().vkrh_last_error().decode())

    def pin_randoms(self, angle_jitter=0.0, gtao_frame_count=0, ssr_counter=0):
        """angle_offset = table[frame_count % 12]/360 + jitter (gtao.cpp:109-111); SURVEY 8(d) pins 60/360 + 0."""
        self._check(lib().vkrh_pin_randoms(self.h, angle_jitter, gtao_frame_count, ssr_counter))

    def set_camera(self, view, prev_view, proj, fazz):
        cam = HostCamera()
        cam.view, cam.prev_view, cam.projection = _mat16(view), _mat16(prev_view), _mat16(proj)
        cam.fovy, cam.aspect, cam.znear, cam.zfar = [float(v) for v in fazz]
        self._check(lib().vkrh_set_camera(self.h, C.byref(cam)))

    def load_scene(self, sc):
        """scene.Scene -> scene::CompiledScene + SceneRenderer inside the frame (main.cpp:250-259)."""
        verts = np.ascontiguousarray(sc.vertices, dtype=np.float32)
        idx = np.ascontiguousarray(sc.indices, dtype=np.uint32)
        draws = (SceneDraw * max(1, len(sc.draws)))()
        for i, d in enumerate(sc.draws):
            m = sc.transforms[d["transform"]][0]
            draws[i] = SceneDraw(_mat16(m), d["vertex_offset"], d["index_offset"], d["index_count"], d["albedo"], d["mr"], 1 if d["flags"] else 0)
        tex = (SceneTexture * max(1, len(sc.textures)))()
        keep = []
        for i, levels in enumerate(sc.textures):
            h, w = levels[0].shape[:2]
            tex[i].width, tex[i].height, tex[i].mip_levels = w, h, len(levels)
            for m, lv in enumerate(levels):
                a = np.ascontiguousarray(lv)
                keep.append(a)
                tex[i].levels[m] = a.ctypes.data
        self._check(lib().vkrh_load_scene(self.h, C.c_void_p(verts.ctypes.data), len(verts), C.c_void_p(idx.ctypes.data), len(idx),
                                          draws, len(sc.draws), tex, len(sc.textures)))

    def set_gathered_mips(self, n):
        """tiled frames: how many whole-frame Hi-Z mips (image mips 1..n) arrive by all-gather; STAGE_HIZ_TAIL rebuilds the rest"""
        self._check(lib().vkrh_set_gathered_mips(self.h, n))

    def pin_screen_trace(self, angle_jitter=0.0, random_offset=0.25, frame_count=0):
        self._check(lib().vkrh_pin_screen_trace(self.h, angle_jitter, random_offset, frame_count))

    def run(self, mask):
        self._check(lib().vkrh_run(self.h, mask))

    def end_frame(self, swap_depth=False):
        self._check(lib().vkrh_end_frame(self.h, 1 if swap_depth else 0))

    def image(self, name, base_mip=0, mip_count=0):
        d = abi.VkrImg()
        self._check(lib().vkrh_image(self.h, name.encode(), base_mip, mip_count, C.byref(d)))
        return d

    CAPTURE_DEPTH_CSV, CAPTURE_DEPTH_PNG, CAPTURE_RGBA_PNG = 0, 1, 2

    def capture(self, name, path, kind, mip=0):
        """ReadBackSystem + the capture writers of main.cpp:118-176."""
        self._check(lib().vkrh_capture(self.h, name.encode(), mip, kind, str(path).encode()))

    def read_buffer(self, name, max_bytes=1 << 24):
        """int32 contents of a named device buffer of the SSR pass (tile lists, indirect arguments)."""
        dst = np.zeros(max_bytes // 4, dtype=np.int32)
        n = C.c_uint64(0)
        self._check(lib().vkrh_read_buffer(self.h, name.encode(), C.c_void_p(dst.ctypes.data), dst.nbytes, C.byref(n)))
        return dst[: n.value // 4]

    def enable_task_timing(self, on=True, only=None):
        """HIP events around every task of each run() — or, with `only`, around the task of that name alone: an event
        pair costs ~3.5 us of queue time, so timing all nine passes slows a 1 ms frame by 7 %."""
        if on and only:
            self._check(lib().vkrh_enable_task_timing_only(self.h, only.encode()))
        else:
            self._check(lib().vkrh_enable_task_timing(self.h, 1 if on else 0))

    def collect_task_times(self):
        """{task name: (total_ms, launches)} measured with HIP events on the frame's stream since the last call."""
        txt = lib().vkrh_collect_task_times(self.h)
        if txt is None:
            raise RuntimeError("host frame error: " + lib().vkrh_last_error().decode())
        out = {}
        for line in txt.decode().splitlines():
            name, ms, n = line.rsplit(" ", 2)
            out[name] = (float(ms), int(n))
        return out

    def last_tasks(self):
        return lib().vkrh_last_tasks(self.h).decode().split()

    def last_lanes(self):
        """stream lane (0 = the frame's stream) of each task of the last run()"""
        return [int(v) for v in lib().vkrh_last_lanes(self.h).decode().split()]

    def set_async(self, on):
        """False (default): one in-order stream.  True: independent passes of one run() overlap on up to three streams."""
        self._check(lib().vkrh_set_async(self.h, 1 if on else 0))

    def download(self, name, layer=None):
        """Copies the named image (or one layer of an array image) into a host ImageBuf with the same layout rules (tests)."""
        from .images import ImageBuf

        if layer is None:
            d = self.image(name)
        else:
            d = abi.VkrImg()
            self._check(lib().vkrh_image_layer(self.h, name.encode(), layer, C.byref(d)))
        buf = ImageBuf(d.format, d.width, d.height, d.mip_count, full=(d.full_width, d.full_height), origin=(d.origin_x, d.origin_y))
        for i in range(d.mip_count):
            assert buf.pitch[i] == d.pitch_bytes[i] and buf.offset[i] == d.mip_offset[i], "layout rules diverged"
        t, off = self.allocator.tensor_at(d.base)
        buf.upload(t[off: off + buf.nbytes].cpu().numpy())
        return buf

    def upload(self, name, host_bytes):
        import torch

        d = self.image(name)
        t, off = self.allocator.tensor_at(d.base)
        src = torch.from_numpy(np.ascontiguousarray(host_bytes, dtype=np.uint8).reshape(-1))
        t[off: off + src.numel()].copy_(src)

    # ---- the C++ tiled frame (native_tiled) ---------------------------------------------------------------
    def tiled_step(self):
        self._check(lib().vkrh_tiled_step(self.tiled_handle))

    def tiled_flush(self):
        self._check(lib().vkrh_tiled_flush(self.tiled_handle))

    def tiled_phase(self, p):
        self._check(lib().vkrh_tiled_phase(self.tiled_handle, p))

    def tiled_time_waits(self, on=True):
        self._check(lib().vkrh_tiled_time_waits(self.tiled_handle, 1 if on else 0))

    def tiled_wait_times(self):
        """{exchange: ms the compute stream stood still for it since the last call} (frame.hpp vkrh_tiled_wait_times)"""
        out = (C.c_float * 5)()
        self._check(lib().vkrh_tiled_wait_times(self.tiled_handle, out))
        return dict(zip(("hiz_gather", "albedo_gather", "taa_halo", "ao_halo", "ssr_halo"), [float(v) for v in out]))

    def tiled_gather_parts(self, which):
        """[(send address, recv address, bytes)] of all-gather `which` (0: Hi-Z mips + normals, 1: albedo)"""
        out = (GatherPart * 8)()
        n = C.c_uint32(0)
        self._check(lib().vkrh_tiled_gather_parts(self.tiled_handle, which, out, 8, C.byref(n)))
        return [(out[i].send, out[i].recv, out[i].bytes) for i in range(n.value)]

    def tiled_halo_peers(self, surface):
        """[(peer rank, send address, recv address, bytes)] of halo refresh `surface` (0: TAA, 1: AO, 2: SSR)"""
        out = (HaloPeer * 2)()
        n = C.c_uint32(0)
        self._check(lib().vkrh_tiled_halo_peers(self.tiled_handle, surface, out, 2, C.byref(n)))
        return [(out[i].peer, out[i].send, out[i].recv, out[i].send_bytes) for i in range(n.value)]

    # the hit-colour request / reply of the C++ tiled frame, step by step (lockstep harness; frame.hpp)
    def tiled_hit_counts(self, world):
        row = (C.c_uint32 * world)()
        self._check(lib().vkrh_tiled_hit_counts(self.tiled_handle, row))
        return list(row)

    def _peer_list(self, out, n):
        return [(out[i].peer, out[i].send, out[i].send_bytes, out[i].recv, out[i].recv_bytes) for i in range(n.value)]

    def tiled_hit_requests(self, matrix):
        """matrix: world x world counts, row-major [requester][owner] -> [(peer, send address, send bytes, recv address, recv bytes)]"""
        flat = (C.c_uint32 * len(matrix))(*matrix)
        out, n = (HaloPeer * 16)(), C.c_uint32(0)
        self._check(lib().vkrh_tiled_hit_requests(self.tiled_handle, flat, out, 16, C.byref(n)))
        return self._peer_list(out, n)

    def tiled_hit_replies(self):
        out, n = (HaloPeer * 16)(), C.c_uint32(0)
        self._check(lib().vkrh_tiled_hit_replies(self.tiled_handle, out, 16, C.byref(n)))
        return self._peer_list(out, n)

    def tiled_hit_finish(self):
        self._check(lib().vkrh_tiled_hit_finish(self.tiled_handle))

    def tiled_hit_errors(self):
        e = C.c_uint32(0)
        self._check(lib().vkrh_tiled_hit_errors(self.tiled_handle, C.byref(e)))
        return int(e.value)

    def tiled_hit_rounds(self):
        """(rounds enqueued on the previous frame's capacities, exact rounds after a host round trip, rounds repeated after an overflow)"""
        r = (C.c_uint64 * 3)()
        self._check(lib().vkrh_tiled_hit_rounds(self.tiled_handle, r))
        return int(r[0]), int(r[1]), int(r[2])

    def tiled_pipelined(self):
        lib().vkrh_tiled_pipelined.argtypes = [C.c_void_p]
        return bool(lib().vkrh_tiled_pipelined(self.tiled_handle))

    def tiled_local_first(self):
        return bool(lib().vkrh_tiled_local_first(self.tiled_handle))

    def tiled_emulate_wire(self, comm_handle, counts):
        """frame.hpp vkrh_tiled_emulate_wire: the harness-driven frame goes on natively on an emulated communicator"""
        flat = (C.c_uint32 * len(counts))(*counts) if counts is not None else None
        self._check(lib().vkrh_tiled_emulate_wire(self.tiled_handle, comm_handle, flat))

    def tiled_hit_bytes(self):
        b = C.c_uint64(0)
        self._check(lib().vkrh_tiled_hit_bytes(self.tiled_handle, C.byref(b)))
        return int(b.value)

    def close(self):
        if self.tiled_handle:
            lib().vkrh_tiled_destroy(self.tiled_handle)
            self.tiled_handle, self.h = None, None
        if self.h:
            lib().vkrh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
