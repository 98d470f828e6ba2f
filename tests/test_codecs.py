"""Storage-format codecs: the oracle's definitions vs the cheap exact forms the HIP kernels use."""
import ctypes as C

import numpy as np
import pytest

from vk_renderer_amd import abi


def test_unorm_decode_fma_form_is_exact():
    """vkr_device.hpp decodes UNORM as fmaf(x, C, x) with x = k * 2^-b and C = nextafter(2^-24) /
    2^-16 + 2^-32 / float(1/255); the oracle defines it as the correctly rounded fp32 quotient
    k / (2^b - 1).  They agree for every code of every width on the path.  (x * C and x + x*C are
    exact in float64, so one rounding to float32 reproduces fmaf.)"""
    consts = {24: float.fromhex("0x1.000002p-24"), 16: float.fromhex("0x1.0001p-16"), 8: float.fromhex("0x1.010102p-8")}
    for bits, c in consts.items():
        assert float(np.float32(c)) == c, "constant must be an exact float32"
        k = np.arange(1 << bits, dtype=np.uint32)
        d = np.float32((1 << bits) - 1)
        ref = (k.astype(np.float32) / d).astype(np.float32)
        x = k.astype(np.float64) * 2.0 ** -bits
        fast = (x + x * c).astype(np.float32)
        assert np.array_equal(ref, fast), f"UNORM{bits}"


@pytest.mark.gpu
def test_device_division_and_decodes_are_exact():
    """On the GPU: div_normal (the kernels' normal-range division) against IEEE '/' for linearize_depth2 over
    all 2^24 stored depths and 2^24 hashed operand pairs, and the three UNORM decodes against their quotients."""
    import torch

    lib = abi.product()
    lib.vkr_selftest_division.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    for znear, zfar in ((0.05, 80.0), (0.1, 1000.0), (1.0, 50.0)):
        counters = torch.zeros(5, dtype=torch.int32, device="cuda")
        abi.check(lib.vkr_selftest_division(counters.data_ptr(), znear, zfar, torch.cuda.current_stream().cuda_stream), lib)
        torch.cuda.synchronize()
        assert counters.tolist() == [0, 0, 0, 0, 0], (znear, zfar, counters.tolist())
    # screen_uv = (g + 0.5) / size of every pixel centre of every extent up to 16384
    lib.vkr_selftest_pixel_uv.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    abi.check(lib.vkr_selftest_pixel_uv(counter.data_ptr(), 16384, torch.cuda.current_stream().cuda_stream), lib)
    torch.cuda.synchronize()
    assert counter.item() == 0


@pytest.mark.gpu
def test_device_square_root_and_reciprocal_are_exact():
    """On the GPU, exhaustively: vkr_device.hpp sqrt_ieee (v_sqrt_f32 + two exact residuals) against sqrtf for EVERY float in
    [2^-96, FLT_MAX], and normalize()'s reciprocal against 1.0f / s for every float in [2^-48, 2^64] — the two correctly
    rounded operations of the numeric contract on the forms the kernels execute."""
    import torch

    lib = abi.product()
    lib.vkr_selftest_sqrt.argtypes = [C.c_void_p, C.c_void_p]
    counters = torch.zeros(2, dtype=torch.int32, device="cuda")
    abi.check(lib.vkr_selftest_sqrt(counters.data_ptr(), torch.cuda.current_stream().cuda_stream), lib)
    torch.cuda.synchronize()
    assert counters.tolist() == [0, 0], counters.tolist()


def test_half_roundtrip_all_codes(oracle_lib):
    lib = oracle_lib
    lib.vkr_ref_half_to_float.restype = C.c_float
    lib.vkr_ref_half_to_float.argtypes = [C.c_uint16]
    lib.vkr_ref_float_to_half.restype = C.c_uint16
    lib.vkr_ref_float_to_half.argtypes = [C.c_float]
    codes = np.arange(1 << 16, dtype=np.uint16)
    ref = codes.view(np.float16).astype(np.float32)
    for c in list(range(0, 1 << 16, 97)) + [0x0001, 0x03FF, 0x0400, 0x7BFF, 0x7C00, 0xFC00, 0x8000]:
        f = lib.vkr_ref_half_to_float(c)
        if np.isnan(ref[c]):
            assert np.isnan(f)
            continue
        assert f == ref[c]
        assert lib.vkr_ref_float_to_half(f) == c


def test_float_to_half_rounds_to_nearest_even(oracle_lib):
    lib = oracle_lib
    lib.vkr_ref_float_to_half.restype = C.c_uint16
    lib.vkr_ref_float_to_half.argtypes = [C.c_float]
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        rng.uniform(-70000, 70000, 2000), rng.uniform(-1, 1, 2000), rng.uniform(-1e-4, 1e-4, 2000), rng.uniform(-1e-7, 1e-7, 500),
        np.array([65504.0, 65519.9, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 0.0, -0.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11])
    ]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    for v, w in zip(vals, want):
        assert lib.vkr_ref_float_to_half(float(v)) == int(w), (v, hex(int(w)))


def test_srgb_tables_match_generated_device_table(oracle_lib):
    """csrc/srgb_tables.inc (device) and the oracle's table are the same 256 floats, and encode inverts decode."""
    import os
    import re
    import struct

    lib = oracle_lib
    lib.vkr_ref_srgb8_to_float.restype = C.c_float
    lib.vkr_ref_srgb8_to_float.argtypes = [C.c_uint8]
    lib.vkr_ref_float_to_srgb8.restype = C.c_uint8
    lib.vkr_ref_float_to_srgb8.argtypes = [C.c_float]
    txt = open(os.path.join(abi.ROOT, "vk-renderer_amd", "csrc", "srgb_tables.inc")).read()
    dec = re.search(r"k_srgb_decode_bits\[256\] = \{(.*?)\};", txt, re.S).group(1)
    bits = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", dec)]
    assert len(bits) == 256
    for i, b in enumerate(bits):
        f = struct.unpack("<f", struct.pack("<I", b))[0]
        assert lib.vkr_ref_srgb8_to_float(i) == f
        assert lib.vkr_ref_float_to_srgb8(f) == i
    assert lib.vkr_ref_float_to_srgb8(-1.0) == 0 and lib.vkr_ref_float_to_srgb8(2.0) == 255
    assert lib.vkr_ref_float_to_srgb8(float("nan")) == 0
