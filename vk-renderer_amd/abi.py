"""ctypes mirror of include/vkr_postfx.h (the C-ABI of the HIP hot path).

Struct layouts must stay byte-identical to the header; tests/test_abi.py checks the
sizes against values the shared library reports and that every declared symbol is
exported.  Nothing here computes anything: it only declares types and loads libraries.
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# VKR_POSTFX_LIB: development only (tools/blur_timeline.py loads an instrumented build of the same sources)
PRODUCT_LIB = os.environ.get("VKR_POSTFX_LIB") or os.path.join(ROOT, "vk-renderer_amd", "csrc", "libvkr_postfx.so")
# VKR_HOST_LIB: tests only (tests/test_reference_passes_gpu.py loads the mirror built around the reference's own pass sources)
HOST_LIB = os.environ.get("VKR_HOST_LIB") or os.path.join(ROOT, "vk-renderer_amd", "host", "libvkr_host.so")

VKR_MAX_MIPS = 16
HALTON_SEQ_SIZE = 128

# vkr_format
FMT_D24_UNORM_S8 = 1
FMT_RG16_UNORM = 2
FMT_RG16_SFLOAT = 3
FMT_RGBA8_SRGB = 4
FMT_RGBA8_UNORM = 5
FMT_RGBA16_UNORM = 6
FMT_RGBA16_SFLOAT = 7
FMT_R16_SFLOAT = 8
FMT_R32_SFLOAT = 9
FMT_R8_UNORM = 10
FMT_RGBA32_SFLOAT = 11
FORMAT_BYTES = {1: 4, 2: 4, 3: 4, 4: 4, 5: 4, 6: 8, 7: 8, 8: 2, 9: 4, 10: 1, 11: 16}

NORMALIZE_REFLECTIONS = 1
ACCUMULATE_REFLECTIONS = 2
BILATERAL_FILTER = 4
SYNTH_DEPTH_ONLY = 1
SYNTH_TEXTURED_ROUGHNESS = 2


class VkrImg(C.Structure):
    _fields_ = [
        ("base", C.c_void_p),
        ("format", C.c_uint32),
        ("mip_count", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("full_width", C.c_uint32),
        ("full_height", C.c_uint32),
        ("origin_x", C.c_int32),
        ("origin_y", C.c_int32),
        ("pitch_bytes", C.c_uint32 * VKR_MAX_MIPS),
        ("mip_offset", C.c_uint64 * VKR_MAX_MIPS),
    ]


class Mat4(C.Structure):
    _fields_ = [("m", C.c_float * 16)]

    @staticmethod
    def from_np(a):
        """a: 4x4 numpy array in maths (row, col) convention -> column-major storage."""
        import numpy as np

        m = Mat4()
        flat = np.asarray(a, dtype=np.float32).T.reshape(-1)
        for i in range(16):
            m.m[i] = float(flat[i])
        return m


class GtaoParams(C.Structure):
    _fields_ = [("normal_mat", Mat4), ("fovy", C.c_float), ("aspect", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class GtaoPush(C.Structure):
    _fields_ = [("angle_offset", C.c_float), ("weight_ratio", C.c_float), ("use_mis", C.c_uint32),
                ("two_directions", C.c_uint32), ("reflections_only", C.c_uint32)]


class GtaoFilterPush(C.Structure):
    _fields_ = [("znear", C.c_float), ("zfar", C.c_float)]


class GtaoAccumParams(C.Structure):
    _fields_ = [("inverse_camera", Mat4), ("prev_inverse_camera", Mat4), ("mvp", Mat4),
                ("fovy_aspect_znear_zfar", C.c_float * 4)]


class GtaoAccumPush(C.Structure):
    _fields_ = [("clear_history", C.c_uint32)]


class TraceParams(C.Structure):
    _fields_ = [("normal_mat", Mat4), ("frame_random", C.c_uint32), ("fovy", C.c_float), ("aspect", C.c_float),
                ("znear", C.c_float), ("zfar", C.c_float)]


class TracePush(C.Structure):
    _fields_ = [("max_roughness", C.c_float)]


class FilterPush(C.Structure):
    _fields_ = [("render_flags", C.c_uint32)]


class BlurPush(C.Structure):
    _fields_ = [("max_roughness", C.c_float), ("accumulate", C.c_uint32), ("disable_blur", C.c_uint32)]


class ReprojectParams(C.Structure):
    _fields_ = [("inverse_camera", Mat4), ("prev_inverse_camera", Mat4), ("fovy_aspect_znear_zfar", C.c_float * 4)]


class SsrParams(C.Structure):
    _fields_ = [("normal_mat", Mat4), ("fovy", C.c_float), ("aspect", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class ShadingParams(C.Structure):
    _fields_ = [("inverse_camera", Mat4), ("camera", Mat4), ("shadow_mvp", Mat4), ("fovy", C.c_float), ("aspect", C.c_float),
                ("znear", C.c_float), ("zfar", C.c_float)]


class ShadingPush(C.Structure):
    _fields_ = [("min_max_roughness", C.c_float * 2), ("show_ao", C.c_uint32)]


class ClassificationPush(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_roughness", C.c_float), ("glossy_value", C.c_float)]


class TraceIndirectPush(C.Structure):
    _fields_ = [("reflection_type", C.c_uint32), ("max_roughness", C.c_float)]


class GtaoGfxPush(C.Structure):
    _fields_ = [("angle_offset", C.c_float)]


class GtaoReprojection(C.Structure):
    _fields_ = [("camera_to_prev_frame", Mat4), ("fovy", C.c_float), ("aspect", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class DeinterleavePush(C.Structure):
    _fields_ = [("pattern_step", C.c_int32)]


class GtaoDeinterleavedPush(C.Structure):
    _fields_ = [("pattern_n", C.c_int32), ("layer", C.c_uint32), ("angle_offset", C.c_float)]


class ScreenTraceParams(C.Structure):
    _fields_ = [("normal_mat", Mat4), ("random_offset", C.c_float), ("angle_offset", C.c_float), ("fovy", C.c_float),
                ("aspect", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class ScreenTraceFilterPush(C.Structure):
    _fields_ = [("znear", C.c_float), ("zfar", C.c_float)]


class ScreenTraceAccumPush(C.Structure):
    _fields_ = [("fovy", C.c_float), ("aspect", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class RasterTransform(C.Structure):
    _fields_ = [("model", Mat4), ("normal", Mat4)]


class RasterDraw(C.Structure):
    _fields_ = [("transform_index", C.c_uint32), ("albedo_index", C.c_uint32), ("mr_index", C.c_uint32), ("flags", C.c_uint32),
                ("index_offset", C.c_uint32), ("index_count", C.c_uint32), ("vertex_offset", C.c_uint32), ("reserved", C.c_uint32)]


class GbufConst(C.Structure):
    _fields_ = [("view_projection", Mat4), ("prev_view_projection", Mat4), ("jitter", C.c_float * 4),
                ("fovy_aspect_znear_zfar", C.c_float * 4)]


class RasterScene(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_uint32), ("indices", C.c_void_p), ("index_count", C.c_uint32),
                ("transforms", C.c_void_p), ("transform_count", C.c_uint32), ("draws", C.c_void_p), ("draw_count", C.c_uint32),
                ("textures", C.c_void_p), ("texture_count", C.c_uint32)]


class SynthParams(C.Structure):
    _fields_ = [("camera_to_world", Mat4), ("prev_mvp", Mat4), ("mvp", Mat4), ("fovy", C.c_float), ("aspect", C.c_float),
                ("znear", C.c_float), ("zfar", C.c_float), ("seed", C.c_uint32), ("flags", C.c_uint32)]


P = C.POINTER
_IMG = P(VkrImg)


class TraceWindowPush(C.Structure):  # vkr_trace_window_push
    _fields_ = [("max_roughness", C.c_float), ("normal_row0", C.c_uint32), ("normal_row1", C.c_uint32)]


class HitSources(C.Structure):  # vkr_hit_sources
    _fields_ = [("rays", _IMG), ("albedo_width", C.c_uint32), ("albedo_height", C.c_uint32), ("window_row0", C.c_uint32),
                ("window_row1", C.c_uint32), ("pending_mask", _IMG), ("pending_data", _IMG),
                ("normal_width", C.c_uint32), ("normal_height", C.c_uint32), ("normal_row0", C.c_uint32), ("normal_row1", C.c_uint32)]


HIT_BOTH_ROWS, HIT_NORMAL, HIT_REPLY_BYTES = 0x10000000, 0x20000000, 16
HIT_WORKSPACE_WORDS = 4096  # include/vkr_postfx.h VKR_HIT_WORKSPACE_WORDS

# name -> argument types *without* the trailing stream
ENTRY_ARGS = {
    "downsample_gbuffer": [_IMG, _IMG, _IMG, _IMG, _IMG],
    "depth_mips": [_IMG, C.c_uint32],
    "pdf_preintegrate": [_IMG],
    "sssr_trace": [_IMG, _IMG, _IMG, P(TraceParams), C.c_void_p, _IMG, _IMG, _IMG, P(TracePush)],
    # the same program in two launches (head + resume over a frame-wide queue of parked rays): workspace, bytes, park_after_rounds
    "sssr_trace_split": [_IMG, _IMG, _IMG, P(TraceParams), C.c_void_p, _IMG, _IMG, _IMG, P(TracePush), C.c_void_p, C.c_uint64, C.c_uint32],
    "sssr_filter": [_IMG, _IMG, _IMG, _IMG, _IMG, _IMG, P(TraceParams), P(FilterPush)],
    "sssr_blur": [_IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, P(ReprojectParams), P(BlurPush)],
    "gtao_main": [_IMG, P(GtaoParams), _IMG, _IMG, _IMG, _IMG, P(GtaoPush)],
    "gtao_filter": [_IMG, _IMG, _IMG, P(GtaoFilterPush)],
    "gtao_accumulate": [_IMG, _IMG, _IMG, _IMG, _IMG, _IMG, P(GtaoAccumParams), P(GtaoAccumPush)],
    "taa_resolve": [_IMG, _IMG, _IMG, _IMG, _IMG, _IMG, P(ReprojectParams)],
    "synth_gbuffer": [_IMG, _IMG, _IMG, _IMG, _IMG, P(SynthParams)],
    "ssr": [_IMG, _IMG, _IMG, P(SsrParams), _IMG, _IMG],
    "brdf_preintegrate": [C.c_void_p, _IMG],
    "defered_shading": [_IMG, _IMG, _IMG, _IMG, P(ShadingParams), _IMG, _IMG, _IMG, _IMG, P(ShadingPush)],
    # G-buffer raster stage; trailing (scratch pointer, scratch bytes)
    "raster_gbuffer": [P(RasterScene), P(GbufConst), _IMG, _IMG, _IMG, _IMG, _IMG, C.c_void_p, C.c_uint64],
    # tile-classified trace: tile lists / indirect args are raw pointers
    "sssr_clear_indirect": [C.c_void_p, C.c_void_p],
    "sssr_classification": [_IMG, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(ClassificationPush)],
    "sssr_trace_indirect": [_IMG, _IMG, _IMG, P(TraceParams), C.c_void_p, _IMG, C.c_void_p, C.c_void_p, C.c_uint32, P(TraceIndirectPush)],
    # dormant GTAO variants; `layers` = array of per-layer descriptors + count
    "gtao_main_graphics": [_IMG, P(GtaoParams), _IMG, _IMG, P(GtaoGfxPush)],
    "gtao_reproject": [P(GtaoReprojection), _IMG, _IMG, _IMG, _IMG, _IMG],
    "deinterleave_depth": [_IMG, _IMG, C.c_uint32, P(DeinterleavePush)],
    "gtao_main_deinterleaved": [_IMG, C.c_uint32, P(GtaoParams), _IMG, _IMG, P(GtaoDeinterleavedPush)],
    # ScreenSpaceTrace (row R2)
    "screen_trace_main": [_IMG, _IMG, _IMG, _IMG, _IMG, P(ScreenTraceParams)],
    "screen_trace_filter": [_IMG, _IMG, _IMG, P(ScreenTraceFilterPush)],
    "screen_trace_accumulate": [_IMG, _IMG, _IMG, _IMG, P(ScreenTraceAccumPush)],
    # multi-GPU: hit colours / hit normals by request / reply (no reference counterpart)
    "sssr_trace_windowed": [_IMG, _IMG, _IMG, P(TraceParams), C.c_void_p, _IMG, _IMG, _IMG, _IMG, _IMG, P(TraceWindowPush)],
    # ... in two launches around the arrival of the whole-frame pyramid: local pyramid first; trailing (workspace, bytes[, park_after_rounds])
    "sssr_trace_windowed_head": [_IMG, _IMG, _IMG, _IMG, P(TraceParams), C.c_void_p, _IMG, _IMG, _IMG, _IMG, _IMG, P(TraceWindowPush), C.c_void_p, C.c_uint64, C.c_uint32],
    "sssr_trace_windowed_resume": [_IMG, _IMG, _IMG, P(TraceParams), C.c_void_p, _IMG, _IMG, _IMG, _IMG, _IMG, P(TraceWindowPush), C.c_void_p, C.c_uint64],
    "sssr_validate": [_IMG, _IMG, _IMG, _IMG, P(TraceParams)],
    "hit_requests": [P(HitSources), P(C.c_uint32), C.c_uint32, C.c_void_p, C.c_void_p, P(C.c_uint32), C.c_void_p],
    # pass 2 into segments of fixed room (workspace, segments, capacities, out, dropped flag); which requests an overflowing
    # segment keeps is the order of the device's atomics: checked by its properties, not against a twin
    "hit_requests_bounded": [P(HitSources), P(C.c_uint32), C.c_uint32, C.c_void_p, P(C.c_uint32), P(C.c_uint32), C.c_void_p, C.c_void_p],
    "hit_reply": [_IMG, _IMG, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p],
    "hit_scatter": [_IMG, _IMG, C.c_void_p, C.c_void_p, C.c_uint32],
}


# entries without a twin in a checker library: a different SCHEDULE of another entry (same images bit for bit), or a variant
# whose result is only defined up to the order of the device's atomics (checked against the exact entry by its properties)
SCHEDULE_VARIANTS = {"sssr_trace_split": "sssr_trace", "sssr_trace_windowed_head": "sssr_trace_windowed", "sssr_trace_windowed_resume": "sssr_trace_windowed",
                     "hit_requests_bounded": "hit_requests"}


class RectCopy(C.Structure):  # vkr_rect_copy
    _fields_ = [("src", C.c_uint64), ("dst", C.c_uint64), ("src_pitch", C.c_uint32), ("dst_pitch", C.c_uint32),
                ("row_bytes", C.c_uint32), ("rows", C.c_uint32)]


class ExtensionMissing(RuntimeError):
    pass


def _load(path, what, needs_hip=True):
    if needs_hip:
        # torch ships its own libamdhip64.so.7; load it first so the extension binds to the same HIP
        # runtime torch uses for memory and streams (two runtimes in one process do not share devices)
        import torch  # noqa: F401
    if not os.path.exists(path):
        raise ExtensionMissing(
            f"{what} not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` first."
        )
    return C.CDLL(path)


_product = None


def product():
    """The HIP C-ABI library.  Fails loudly when it is missing — there is no fallback."""
    global _product
    if _product is None:
        lib = _load(PRODUCT_LIB, "HIP extension libvkr_postfx.so")
        for name, args in ENTRY_ARGS.items():
            fn = getattr(lib, "vkr_" + name)
            fn.argtypes = args + [C.c_void_p]
            fn.restype = C.c_int
        lib.vkr_stream_read.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.vkr_stream_read.restype = C.c_int
        lib.vkr_copy_rects.argtypes = [P(RectCopy), C.c_uint32, C.c_void_p]
        lib.vkr_comm_unique_id.argtypes = [C.c_void_p]
        lib.vkr_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_void_p)]
        lib.vkr_comm_destroy.argtypes = [C.c_void_p]
        lib.vkr_comm_create_emulated.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, P(C.c_void_p)]
        lib.vkr_comm_create_emulated.restype = C.c_int
        lib.vkr_comm_available.argtypes = []
        lib.vkr_get_switches.argtypes = []
        lib.vkr_get_switches.restype = C.c_uint32
        lib.vkr_set_switches.argtypes = [C.c_uint32]
        lib.vkr_set_switches.restype = None
        lib.vkr_comm_available.restype = C.c_int
        lib.vkr_copy_rects.restype = C.c_int
        lib.vkr_sssr_trace_workspace_bytes.argtypes = [C.c_uint32, C.c_uint32]
        lib.vkr_sssr_trace_workspace_bytes.restype = C.c_uint64
        lib.vkr_raster_scratch_bytes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        lib.vkr_raster_scratch_bytes.restype = C.c_uint64
        lib.vkr_halton23_fill.argtypes = [C.c_void_p, C.c_uint32]
        lib.vkr_halton23_fill.restype = None
        lib.vkr_version.restype = C.c_char_p
        lib.vkr_last_error.restype = C.c_char_p
        lib.vkr_format_bytes.argtypes = [C.c_uint32]
        lib.vkr_format_bytes.restype = C.c_uint32
        _product = lib
    return _product


def check(rc, lib=None):
    if rc != 0:
        msg = ""
        if lib is not None and hasattr(lib, "vkr_last_error"):
            msg = (lib.vkr_last_error() or b"").decode()
        raise RuntimeError(f"vkr call failed with code {rc}: {msg}")


COMM_ID_BYTES = 128
# measurement switches (include/vkr_postfx.h VKR_SWITCH_*): vkr_get_switches / vkr_set_switches
SWITCH_BLUR_NO_SKIP, SWITCH_FILTER_NO_SKIP, SWITCH_TAA_GENERIC, SWITCH_SHADING_GENERIC, SWITCH_BLUR_GENERIC, SWITCH_TRACE_ONE_LAUNCH, SWITCH_BLUR_LANE_LOOPS = 1, 2, 4, 8, 16, 32, 64


class Comm:
    """RCCL communicator of the C-ABI (include/vkr_postfx.h vkr_comm_*): the wire of the multi-GPU frame.  Rank 0 makes
    the id, `share` hands its bytes to every rank out of band (e.g. torch.distributed.broadcast_object_list over gloo),
    then every rank creates the communicator collectively with its device current.

    Every rank takes the same branch whatever fails where (a mismatch would leave the others blocked in a collective):
      * `agree(ok)` — optional, a collective AND over the ranks on the control plane — is asked whether RCCL loads on
        EVERY rank before anything collective happens;
      * every rank always calls `share` (rank 0 with None when it could not make the id) and raises after it;
      * `self_check(agree)` verifies a fresh communicator by moving known bytes through all three exchanges."""

    def __init__(self, rank, world, share, agree=None):
        lib = product()
        self.handle, self.rank, self.world = None, rank, world
        available = lib.vkr_comm_available() == 0
        why = "" if available else (lib.vkr_last_error() or b"").decode()
        if agree is not None and not agree(available):
            raise RuntimeError("RCCL is not available on every rank" + (f" (this rank: {why})" if why else ""))
        ident = None
        if rank == 0:
            buf = (C.c_uint8 * COMM_ID_BYTES)()
            if available and lib.vkr_comm_unique_id(buf) == 0:
                ident = bytes(buf)
            else:
                why = (lib.vkr_last_error() or b"").decode()
        ident = share(ident)  # rank 0 always reaches this, so nobody waits for an id that never comes
        if ident is None:
            raise RuntimeError("rank 0 could not make a communicator id" + (f": {why}" if why else ""))
        if not available:  # only without `agree`: the ranks that do have RCCL are about to block in vkr_comm_create — pass `agree`
            raise RuntimeError(f"RCCL is not available on this rank: {why}")
        assert len(ident) == COMM_ID_BYTES
        h = C.c_void_p(0)
        check(lib.vkr_comm_create((C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(ident), rank, world, C.byref(h)), lib)
        self.handle = h.value

    @classmethod
    def emulated(cls, rank, world, link_gbps=60.0, launch_us=15.0):
        """vkr_comm_create_emulated: moves nothing, holds the stream for the wire's time (tools/wire_emulation.py)"""
        self = cls.__new__(cls)
        lib = product()
        h = C.c_void_p(0)
        check(lib.vkr_comm_create_emulated(rank, world, link_gbps, launch_us, C.byref(h)), lib)
        self.handle = h.value
        return self

    def self_check(self, device, agree=None):
        """Moves rank-tagged patterns through vkr_all_gather, vkr_all_gather_v and vkr_halo_exchange and verifies every
        received byte (vkr_comm_selfcheck).  Returns True when the wire delivers what the tiled frame expects — on
        every rank, if `agree` (collective AND) is given."""
        import torch

        lib = product()
        lib.vkr_comm_selfcheck_bytes.argtypes = [C.c_int]
        lib.vkr_comm_selfcheck_bytes.restype = C.c_uint64
        lib.vkr_comm_selfcheck.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        scratch = torch.empty(int(lib.vkr_comm_selfcheck_bytes(self.world)), dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device)
        ok = lib.vkr_comm_selfcheck(C.c_void_p(self.handle), C.c_void_p(scratch.data_ptr()), C.c_void_p(stream.cuda_stream)) == 0
        self.self_check_error = "" if ok else (lib.vkr_last_error() or b"").decode()
        return agree(ok) if agree is not None else ok

    def close(self):
        if self.handle:
            check(product().vkr_comm_destroy(C.c_void_p(self.handle)), product())
            self.handle = None
