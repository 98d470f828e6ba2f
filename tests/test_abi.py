"""The C-ABI library loads and exports every symbol include/vkr_postfx.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from vk_renderer_amd import abi


def _declared_symbols():
    txt = open(os.path.join(abi.ROOT, "include", "vkr_postfx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vkr_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_entries():
    syms = _declared_symbols()
    for name in abi.ENTRY_ARGS:
        assert "vkr_" + name in syms
    assert {"vkr_version", "vkr_last_error", "vkr_format_bytes", "vkr_stream_read"} <= set(syms)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(abi.PRODUCT_LIB):
        pytest.skip("HIP library not built yet (run __graft_entry__.build())")
    lib = abi._load(abi.PRODUCT_LIB, "HIP extension")
    for sym in _declared_symbols():
        assert hasattr(lib, sym), f"{sym} declared in include/vkr_postfx.h but not exported"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(abi, "PRODUCT_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(abi, "_product", None)
    with pytest.raises(abi.ExtensionMissing):
        abi.product()


def test_struct_sizes_match_header():
    # sizes computed from the C declarations (4-byte fields, 8-byte pointers / uint64)
    assert C.sizeof(abi.VkrImg) == 8 + 4 * 8 + 4 * 16 + 8 * 16
    assert C.sizeof(abi.Mat4) == 64
    assert C.sizeof(abi.GtaoParams) == 80 and C.sizeof(abi.GtaoPush) == 20
    assert C.sizeof(abi.GtaoAccumParams) == 64 * 3 + 16
    assert C.sizeof(abi.TraceParams) == 64 + 20
    assert C.sizeof(abi.ReprojectParams) == 64 * 2 + 16
    assert C.sizeof(abi.SynthParams) == 64 * 3 + 24
    if os.path.exists(abi.PRODUCT_LIB):
        lib = abi.product()
        for fmt, nbytes in abi.FORMAT_BYTES.items():
            assert lib.vkr_format_bytes(fmt) == nbytes


def test_host_library_loads_and_rejects_unknown_program():
    if not os.path.exists(abi.HOST_LIB):
        pytest.skip("host library not built yet")
    from vk_renderer_amd import host

    lib = host.lib()
    for sym in ("vkrh_create", "vkrh_destroy", "vkrh_run", "vkrh_end_frame", "vkrh_image", "vkrh_set_camera", "vkrh_set_allocator",
                "vkrh_collect_task_times", "vkrh_pin_randoms"):
        assert hasattr(lib, sym)


def test_balance_rows_policy():
    """vkrh_balance_rows: equal times keep equal strips; a frame whose lower half costs twice as much per row gets its
    cut where the cumulative cost is half; alignment and the minimum height hold."""
    from vk_renderer_amd import host

    assert host.balance_rows([1.0, 1.0, 1.0, 1.0], [0, 160, 320, 480, 640]) == [0, 160, 320, 480, 640]
    b = host.balance_rows([1.0, 2.0], [0, 1600, 3200])      # cost 1 per 1600 rows above, 2 below: half of 3 at row 1600 + 400
    assert b == [0, 2000, 3200]
    b = host.balance_rows([10.0, 0.1, 0.1, 0.1], [0, 256, 512, 768, 1024], align=16, min_rows=64)
    assert b[0] == 0 and b[-1] == 1024 and all(x % 16 == 0 for x in b) and all(b[i + 1] - b[i] >= 64 for i in range(4))
    assert b[1] == 64  # the expensive strip shrinks to the minimum
    with pytest.raises(RuntimeError):
        host.balance_rows([1.0, 1.0], [0, 64, 100])  # frame height not a multiple of the alignment
