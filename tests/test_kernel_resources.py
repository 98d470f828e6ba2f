"""Build-time guard for the hot kernels: registers, scratch, LDS and the occupancy that follows from them, as the compiler
reports them for gfx950 (tools/kernel_resources.py, `-Rpass-analysis=kernel-resource-usage`; hipcc cross-compiles, no GPU).

Round 3 shipped k_gtao_main with 6 VGPRs in scratch memory (20 B per lane through HBM on the hot path) behind a
`__launch_bounds__` that bought its occupancy; nothing failed.  The limits here are the state of the tree: a kernel that
starts to spill, crosses a register step of the occupancy table or outgrows its share of the 160 KB of LDS fails this test.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import kernel_resources  # noqa: E402

LDS_PER_CU = 160 * 1024
SIMDS_PER_CU = 4
MAX_WAVES_PER_SIMD = 8

# kernel: (source, threads per block, max VGPRs, max scratch bytes per lane, max LDS bytes per block, min resident waves per SIMD)
HOT = {
    "k_downsample_gbuffer": ("hiz.hip", 256, 32, 0, 0, 8),
    "k_depth_mips_fused": ("hiz.hip", 256, 32, 0, 2048, 8),
    # <WINDOWED, PARK, LOCAL>: one launch / head of the split (single GPU) / windowed one launch / multi-GPU head on local rows
    "k_sssr_trace<false, false, false>": ("ssr.hip", 256, 72, 0, 23 * 1024, 7),
    "k_sssr_trace<false, true, false>": ("ssr.hip", 256, 72, 0, 23 * 1024, 7),
    "k_sssr_trace<true, false, false>": ("ssr.hip", 256, 72, 0, 23 * 1024, 7),
    "k_sssr_trace<true, true, true>": ("ssr.hip", 256, 80, 0, 23 * 1024, 6),
    # the resume launch is latency-bound (a tenth of the rays, 3 blocks per CU): registers are not what limits it
    "k_sssr_trace_resume<false, false>": ("ssr.hip", 256, 96, 0, 1024, 5),      # <WINDOWED, COMPACT>: single GPU, every lane its own ray
    "k_sssr_trace_resume<true, true>": ("ssr.hip", 256, 96, 0, 23 * 1024, 5),   # multi-GPU resume: compacted rounds
    "k_sssr_filter": ("ssr.hip", 256, 64, 0, 12 * 1024 + 256, 8),
    "k_sssr_blur": ("ssr.hip", 512, 128, 16, 48 * 1024, 4),
    "k_gtao_main<true, true>": ("gtao.hip", 1024, 64, 0, 21 * 1024, 8),   # 16-wave blocks: > 64 VGPRs means ONE block per CU
    "k_gtao_main<true, false>": ("gtao.hip", 1024, 64, 0, 21 * 1024, 8),
    "k_gtao_main<false, true>": ("gtao.hip", 1024, 64, 0, 2048, 8),
    "k_gtao_main<false, false>": ("gtao.hip", 1024, 64, 0, 2048, 8),
    "k_gtao_filter": ("gtao.hip", 256, 64, 0, 4096, 8),
    "k_gtao_accumulate": ("gtao.hip", 256, 64, 0, 0, 8),
    "k_taa_resolve<true, true>": ("taa.hip", 256, 64, 0, 6 * 1024, 8),
    "k_taa_resolve<true, false>": ("taa.hip", 256, 64, 0, 2048, 8),
    "k_taa_resolve<false, false>": ("taa.hip", 256, 64, 0, 2048, 8),
}


def resident_waves_per_simd(threads, vgprs, lds_bytes):
    """waves per SIMD a CU actually holds: whole blocks only, limited by registers (512 VGPRs per SIMD lane, allocated in
    steps of 8) and by LDS"""
    waves_per_block = (threads + 63) // 64
    alloc = max(8, (vgprs + 7) // 8 * 8)
    by_regs = min(MAX_WAVES_PER_SIMD, 512 // alloc) * SIMDS_PER_CU // waves_per_block
    by_lds = LDS_PER_CU // lds_bytes if lds_bytes else 10 ** 9
    by_slots = MAX_WAVES_PER_SIMD * SIMDS_PER_CU // waves_per_block
    blocks = min(by_regs, by_lds, by_slots)
    return blocks * waves_per_block / SIMDS_PER_CU


@pytest.fixture(scope="module")
def res():
    return kernel_resources.resources(sorted({v[0] for v in HOT.values()}))


def test_every_hot_kernel_is_reported(res):
    missing = [k for k in HOT if k not in res]
    assert not missing, f"kernels not found in the compiler's remarks (renamed?): {missing}; have {sorted(res)}"


@pytest.mark.parametrize("kernel", sorted(HOT))
def test_hot_kernel_resources(res, kernel):
    src, threads, max_vgprs, max_scratch, max_lds, min_waves = HOT[kernel]
    r = res[kernel]
    assert r["file"] == src
    assert r["vgprs"] <= max_vgprs, f"{kernel}: {r['vgprs']} VGPRs > {max_vgprs}"
    assert r["scratch_bytes"] <= max_scratch, f"{kernel}: {r['scratch_bytes']} B of scratch per lane (limit {max_scratch}): a spill on the hot path"
    assert r["lds_bytes"] <= max_lds, f"{kernel}: {r['lds_bytes']} B of LDS per block > {max_lds}"
    waves = resident_waves_per_simd(threads, r["vgprs"], r["lds_bytes"])
    assert waves >= min_waves, f"{kernel}: {waves} resident waves per SIMD < {min_waves} ({r['vgprs']} VGPRs, {r['lds_bytes']} B LDS, {threads} threads per block)"


def test_no_hot_kernel_but_the_blur_touches_scratch(res):
    spilling = {k: res[k]["scratch_bytes"] for k in HOT if res[k]["scratch_bytes"] > 0}
    assert set(spilling) <= {"k_sssr_blur"}, spilling
