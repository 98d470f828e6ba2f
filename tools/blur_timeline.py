#!/usr/bin/env python3
"""Per-block timeline of the SSR blur at 3840x2160 (development tool; run on the GPU box).

Builds csrc/ with -DVKR_BLUR_STAMPS into gpurun_out/dbg/libvkr_postfx.so (wave 0 of every block stamps the 100 MHz wall clock
at: 0 start, 1 tile staged, 2 per-pixel set-up done, 3 tap loop done, 4 end; 5 = blur radius of the block's first pixel,
6 = XCC_ID << 32 | HW_ID), runs one frame of the chain and a few blur launches, and prints: phase durations by radius class,
machine occupancy over time, and what the Sigma(block time) / slots bound would be.

    python tools/blur_timeline.py            # on the GPU box: gpurun -- python tools/blur_timeline.py
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "dbg")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "vk-renderer_amd", "csrc")
    objs = []
    procs = []
    for f in sorted(os.listdir(src)):
        if f.endswith(".hip"):
            o = os.path.join(OUT, f[:-4] + ".o")
            objs.append(o)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                                           "-fno-gpu-flush-denormals-to-zero", "-DVKR_CONTRACT=2", "-DVKR_BLUR_STAMPS", "-Wno-unused-function", "-c", os.path.join(src, f), "-o", o]))
    assert all(p.wait() == 0 for p in procs)
    lib = os.path.join(OUT, "libvkr_postfx.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    return lib


def main():
    lib_path = build()
    os.environ["VKR_POSTFX_LIB"] = lib_path
    import numpy as np
    import torch

    import vk_renderer_amd  # noqa: F401
    from vk_renderer_amd import abi
    from vk_renderer_amd.chain import PostFxChain

    W, H = 3840, 2160
    c = PostFxChain(W, H, backend="product", device="cuda")
    c.synth(); c.build_prev_hiz(); c.preintegrate_pdf()
    c.frame(); c.swap_histories(); c.frame()
    lib = abi.product()
    lib.vkr_debug_blur_stamps.argtypes = [C.c_void_p, C.c_uint64]
    for _ in range(3):
        c.ssr_blur()
    c.sync()
    tiles = ((W // 2 + 31) // 32) * ((H // 2 + 31) // 32)
    st = np.zeros((tiles, 8), dtype=np.uint64)
    abi.check(lib.vkr_debug_blur_stamps(st.ctypes.data_as(C.c_void_p), st.nbytes), lib)
    np.save(os.path.join(ROOT, "gpurun_out", "blur_stamps.npy"), st)
    t0 = st[:, 0].min()
    t = (st[:, :5] - t0).astype(np.float64) * 0.01  # microseconds
    total = t[:, 4].max()
    dur = t[:, 4] - t[:, 0]
    r = st[:, 5].astype(np.int64)
    print(f"kernel {total:.1f} us, {tiles} blocks, sum(block time) / 512 slots = {dur.sum() / 512:.1f} us, mean concurrency {dur.sum() / total:.0f}")
    print("radius   n   stage  setup   loop  store  total   (us, wave 0 of the block)")
    for rr in sorted(set(r.tolist())):
        m = r == rr
        ph = [(t[m, k + 1] - t[m, k]).mean() for k in range(4)]
        print(f"{rr:6d} {m.sum():4d}  {ph[0]:6.1f} {ph[1]:6.1f} {ph[2]:6.1f} {ph[3]:6.1f} {dur[m].mean():6.1f}")
    print("occupancy over time (blocks resident):")
    for x in np.linspace(0, total, 21)[:-1]:
        print(f"  {x:6.1f} us: {int(((t[:, 0] <= x) & (t[:, 4] > x)).sum())}")
    xcc = (st[:, 6] >> np.uint64(32)).astype(np.int64)
    print("blocks per XCC:", np.bincount(xcc & 15)[:8].tolist())
    # Per CU (XCC, shader engine, array, CU of HW_ID): how much of the kernel's time the CU holds 0 / 1 / 2+ blocks, and how
    # much of it EVERY block it holds is still staging its tile (phase 0 -> 1: global loads, decode, LDS stores, barrier) —
    # time in which the CU issues next to no VALU work.  This is what VALU-busy 0.69 is made of.
    hw = st[:, 6] & np.uint64(0xFFFFFFFF)
    cu_key = ((xcc & 15) << 16) | (((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64) << 8) | (((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64) << 4) | ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    cus = np.unique(cu_key)
    grid = np.arange(0.0, total, 0.25)  # 0.25 us resolution
    empty = one = two = staging_only = 0.0
    for k in cus:
        m = cu_key == k
        res = np.zeros(grid.size, dtype=np.int32)
        tapping = np.zeros(grid.size, dtype=np.int32)
        for b in np.flatnonzero(m):
            res += (grid >= t[b, 0]) & (grid < t[b, 4])
            tapping += (grid >= t[b, 1]) & (grid < t[b, 3])
        empty += (res == 0).mean(); one += (res == 1).mean(); two += (res >= 2).mean()
        staging_only += ((res > 0) & (tapping == 0)).mean()
    n = float(len(cus))
    summary = {"kernel_us": total, "blocks": int(tiles), "cus_seen": int(n), "cu_time_share": {"no_block": empty / n, "one_block": one / n, "two_or_more": two / n,
               "every_resident_block_staging_or_storing": staging_only / n},
               "sum_block_time_over_512_slots_us": dur.sum() / 512, "phase_us_by_radius": {}}
    for rr in sorted(set(r.tolist())):
        mm = r == rr
        summary["phase_us_by_radius"][int(rr)] = {"blocks": int(mm.sum()), "stage": float((t[mm, 1] - t[mm, 0]).mean()), "setup": float((t[mm, 2] - t[mm, 1]).mean()),
                                                   "taps": float((t[mm, 3] - t[mm, 2]).mean()), "store": float((t[mm, 4] - t[mm, 3]).mean())}
    print(f"per CU ({int(n)} CUs seen): no block {empty / n:.3f}, one block {one / n:.3f}, two or more {two / n:.3f} of the kernel's time; "
          f"every resident block outside its tap loop {staging_only / n:.3f}")
    import json
    with open(os.path.join(ROOT, "gpurun_out", "blur_timeline.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # start time of blocks by launch order: how far the dispatcher is ahead
    order = np.arange(tiles)
    late = t[:, 0] > 0.6 * total
    print(f"blocks started after 60 % of the kernel: {int(late.sum())}, their mean radius {r[late].mean():.1f}, mean duration {dur[late].mean():.1f} us")


if __name__ == "__main__":
    main()
