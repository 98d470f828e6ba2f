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
