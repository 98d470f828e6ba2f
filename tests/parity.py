"""Comparison rule shared by the parity tests.

north_star: "within 1e-3 relative per pixel".  Outputs live in quantised storage formats
(fp16, UNORM8/16, D24), so a value that sits on a rounding boundary may land one code apart
even when the fp32 results agree to 1e-7.  A texel therefore passes when
    |got - ref| <= REL_TOL * |ref|     or     |got - ref| <= one storage step at that value.
Integer / index outputs (depth pyramid, downsampled normals & velocity) must be bit-exact.
"""
import numpy as np

from vk_renderer_amd import abi

REL_TOL = 1e-3

# every comparison made through report() / record() since the last reset: tests at BASELINE sizes dump this table
# (tests/conftest.py: parity_table fixture -> gpurun_out/parity_<test>.json, copied to profiles/ when judged)
ROWS = []


def record(name, texels, bit_equal, outside_tol, max_abs_diff, rule):
    ROWS.append({"image": name, "texels": int(texels), "bit_equal": int(bit_equal), "outside_tolerance": int(outside_tol),
                 "max_abs_diff": float(max_abs_diff), "rule": rule})


def _srgb_table():
    c = np.arange(256, dtype=np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def storage_step(fmt, ref):
    if fmt in (abi.FMT_RGBA8_UNORM, abi.FMT_R8_UNORM):
        return np.full_like(ref, 1.0 / 255.0)
    if fmt == abi.FMT_RGBA8_SRGB:
        # one stored CODE: in linear light the gap between neighbouring codes grows from 3e-4 (dark) to 8.6e-3 (code 254 -> 255),
        # so "1 / 255" is less than one code for every texel brighter than code 124.  The step at a
        # decoded value is the larger of the gaps to its two neighbours; alpha is stored linearly.
        t = _srgb_table()
        code = np.clip(np.searchsorted(t, ref.astype(np.float64) - 1e-9), 0, 255)
        gap = np.maximum(t[np.minimum(code + 1, 255)] - t[code], t[code] - t[np.maximum(code - 1, 0)])
        step = gap.astype(np.float32)
        if ref.ndim == 3 and ref.shape[-1] == 4:
            step[..., 3] = 1.0 / 255.0
        return step
    if fmt in (abi.FMT_RG16_UNORM, abi.FMT_RGBA16_UNORM):
        return np.full_like(ref, 1.0 / 65535.0)
    if fmt in (abi.FMT_RG16_SFLOAT, abi.FMT_RGBA16_SFLOAT, abi.FMT_R16_SFLOAT):
        a = np.maximum(np.abs(ref), 2.0 ** -14)
        return (2.0 ** (np.floor(np.log2(a)) - 10)).astype(np.float32)
    if fmt == abi.FMT_D24_UNORM_S8:
        return np.full_like(ref, 1.0 / 16777215.0)
    return np.zeros_like(ref)


def mismatches(fmt, got, ref, input_fmt=None):
    """got/ref: decoded float arrays.  Returns boolean array of failing texels (any channel).
    input_fmt: the surface is a blend of an input stored in that (coarser) format whose two versions may differ by one code
    (the TAA target resolves the sRGB8 composite): one storage step of the INPUT at the value is allowed as well."""
    got = got.astype(np.float64)
    ref64 = ref.astype(np.float64)
    both_nan = np.isnan(got) & np.isnan(ref64)
    same_inf = np.isinf(got) & np.isinf(ref64) & (np.sign(got) == np.sign(ref64))
    with np.errstate(invalid="ignore"):
        diff = np.abs(got - ref64)
        ok = (diff <= REL_TOL * np.abs(ref64)) | (diff <= storage_step(fmt, ref.astype(np.float32)).astype(np.float64) * 1.0001)
        if input_fmt is not None:
            ok = ok | (diff <= storage_step(input_fmt, ref.astype(np.float32)).astype(np.float64) * 1.0001)
    ok = ok | both_nan | same_inf
    return ~ok.all(axis=-1)


def report(name, fmt, got, ref):
    bad = mismatches(fmt, got, ref)
    n = int(bad.sum())
    with np.errstate(invalid="ignore"):
        maxabs = float(np.nanmax(np.abs(got.astype(np.float64) - ref.astype(np.float64)))) if got.size else 0.0
    exact = int((got == ref).all(axis=-1).sum()) if got.size else 0
    print(f"[parity] {name:14s} texels {bad.size:9d}  bit-equal {exact:9d}  outside-tol {n:6d}  max|diff| {maxabs:.3e}")
    record(name, bad.size, exact, n, maxabs, f"|d| <= {REL_TOL} |ref| or one storage step")
    return n, bad
