"""The C++ host mirror keeps the reference's error behaviour (exceptions with the reference's messages)
and graph semantics; runs on the CPU with a malloc-backed allocator (no kernel is launched)."""
import ctypes as C
import os

import pytest

from vk_renderer_amd import abi


@pytest.fixture(scope="module")
def report():
    if not os.path.exists(abi.HOST_LIB):
        pytest.skip("host library not built yet")
    from vk_renderer_amd import host

    l = host.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    alloc = host._ALLOC(lambda n, u: libc.malloc(n))
    free = host._FREE(lambda p, u: libc.free(p))
    l.vkrh_set_allocator(alloc, free, None)
    buf = C.create_string_buffer(8192)
    l.vkrh_selftest_errors.argtypes = [C.c_char_p, C.c_uint32]
    assert l.vkrh_selftest_errors(buf, 8192) == 0
    l.vkrh_set_allocator(host._ALLOC(0), host._FREE(0), None)  # back to hipMalloc
    return dict(line.split(": ", 1) for line in buf.value.decode().splitlines())


def test_unknown_program_is_rejected(report):
    assert report["unknown_program"] == "Program not found"  # gpu/shader_program.cpp:197
    assert report["known_program"] == "no error"


def test_downsample_pass_error_messages(report):
    assert report["single_mip_depth"] == "Can't downsample depth texture with 1 mip level"  # downsample_pass.cpp:38
    assert report["mismatched_outputs"] == "Output textures have different sizes"  # downsample_pass.cpp:49


def test_usage_tracking(report):
    assert report["incompatible_usage"] == "Incompatible image usage in task"  # resources.cpp:351
    assert report["read_then_write_in_separate_tasks"] == "no error"


def test_remap_and_ordering(report):
    assert report["remap_keeps_ids_valid"] == "no error"
    assert report["tasks_run_in_submission_order"] == "no error"


def test_misc_errors(report):
    assert "ray-query" in report["ray_query_gtao"]
    assert report["ubo_ring_overflow"] == "Not enough space in uniform buffer"


def test_capture_writers(tmp_path):
    """the capture formats of main.cpp:118-176 — CSV of 24-bit hex depth with the reference's
    exact header / separators, PNG of the masked depth words, PNG of RGBA8 with alpha forced to 255."""
    import numpy as np
    from PIL import Image

    if not os.path.exists(abi.HOST_LIB):
        pytest.skip("host library not built yet")
    from vk_renderer_amd import host

    l = host.lib()
    l.vkrh_selftest_writers.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    W, H = 37, 11
    assert l.vkrh_selftest_writers(str(tmp_path).encode(), W, H) == 0, l.vkrh_last_error().decode()
    x, y = np.meshgrid(np.arange(W, dtype=np.uint64), np.arange(H, dtype=np.uint64))
    depth = ((x * 65537 + y * 257 + 0xAB000000) & 0xFFFFFFFF).astype(np.uint32) & 0xFFFFFF

    lines = (tmp_path / "depth.csv").read_text().split("\n")
    assert lines[0] == "y, " + ",".join(str(i) for i in range(W))  # main.cpp:123-131
    assert lines[-1] == "" and len(lines) == H + 2
    for row in range(H):
        cells = lines[1 + row].split(",")
        assert cells[0] == str(row)
        assert cells[1:] == ["0x%x" % v for v in depth[row]]  # std::hex: lower case, no padding

    png = np.array(Image.open(tmp_path / "depth.png"))
    assert png.shape == (H, W, 4)
    words = png.astype(np.uint32)
    assert np.array_equal(words[..., 0] | (words[..., 1] << 8) | (words[..., 2] << 16) | (words[..., 3] << 24), depth)

    col = np.array(Image.open(tmp_path / "color.png"))
    assert np.array_equal(col[..., 0], (x & 0xFF).astype(np.uint8)) and np.array_equal(col[..., 1], (y & 0xFF).astype(np.uint8))
    assert np.array_equal(col[..., 2], ((x ^ y) & 0xFF).astype(np.uint8)) and np.all(col[..., 3] == 255)


@pytest.mark.gpu
def test_async_lanes_follow_declared_hazards():
    """RenderGraph::set_async(true): the tasks of one submission spread over stream lanes along the hazards of their
    declared accesses — SSR chain on the frame's stream, the GTAO chain (needs the trace's output) on a second lane,
    TAA (needs only the pyramid) on a third — and the frame's outputs stay bit-identical to the one-stream frame."""
    import numpy as np

    from vk_renderer_amd import host
    from vk_renderer_amd.camera import FrameSetup

    outs = {}
    for overlap in (False, True):
        frame = host.HostFrame(FrameSetup(512, 288), device="cuda")
        frame.set_async(overlap)
        frame.run(host.STAGE_LUT | host.STAGE_GBUFFER | host.STAGE_PREV_DEPTH)
        for _ in range(3):
            frame.run(host.STAGE_CHAIN)
            tasks, lanes = frame.last_tasks(), frame.last_lanes()
            frame.end_frame()
        assert tasks == ["DownsampleGbuffer", "DownsampleDepth", "SSSR_trace", "SSSR_filter", "SSSR_blur", "GTAO_main", "GTAO_filter",
                         "GTAO_accumulate", "TAA"]
        assert lanes == ([0, 0, 0, 0, 0, 1, 1, 1, 2] if overlap else [0] * 9)
        outs[overlap] = {n: frame.download(n).to_host().copy() for n in ("rays", "raw", "reflections", "blurred_hist", "filtered", "acc_hist", "taa_hist")}
        frame.close()
    for n in outs[False]:
        assert np.array_equal(outs[False][n], outs[True][n]), f"{n}: overlapped frame differs from the one-stream frame"


def test_null_handles_are_messages_not_crashes():
    """Every vkrh_* entry that takes a frame (or tiled-frame) handle answers NULL with a non-zero status and a message:
    the Python harness keeps None for frames it did not create natively (tools/lockstep_profile.py --world 1 found the
    crash this replaces).  No device call is reached, so this runs without a GPU."""
    import ctypes as C

    from vk_renderer_amd import host

    l = host.lib()
    l.vkrh_last_error.restype = C.c_char_p
    for name, extra in (("vkrh_run", (C.c_uint32(1),)), ("vkrh_end_frame", (C.c_uint32(0),)), ("vkrh_set_async", (C.c_uint32(1),)),
                        ("vkrh_enable_task_timing", (C.c_uint32(1),)), ("vkrh_tiled_step", ()), ("vkrh_tiled_flush", ()),
                        ("vkrh_tiled_phase", (C.c_uint32(0),)), ("vkrh_tiled_time_waits", (C.c_uint32(1),))):
        fn = getattr(l, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p] + [type(e) for e in extra]
        assert fn(None, *extra) != 0, name
        assert b"NULL" in l.vkrh_last_error(), (name, l.vkrh_last_error())
    l.vkrh_collect_task_times.restype = C.c_char_p
    l.vkrh_collect_task_times.argtypes = [C.c_void_p]
    assert l.vkrh_collect_task_times(None) is None
    l.vkrh_last_tasks.restype = C.c_char_p
    l.vkrh_last_tasks.argtypes = [C.c_void_p]
    assert l.vkrh_last_tasks(None) == b""
