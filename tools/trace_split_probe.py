#!/usr/bin/env python3
"""GPU probe of vkr_sssr_trace_split: bit-equality of `rays` / `raw` with the one-launch trace and the time of each variant
(HIP events around 20 launches).  python tools/trace_split_probe.py [W H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from vk_renderer_amd.chain import PostFxChain  # noqa: E402


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    c = PostFxChain(W, H, backend="product", device="cuda")
    c.synth(); c.build_prev_hiz(); c.init_histories(); c.preintegrate_pdf(); c.downsample()
    c.ssr_trace()
    c.sync()
    want_rays, want_raw = c.rays.raw(0).copy(), c.raw.raw(0).copy()

    def timed(split, n=20):
        for _ in range(3):
            c.ssr_trace(split=split)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            c.ssr_trace(split=split)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    print(f"{W}x{H}: one launch {timed(None):.4f} ms")
    only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0, 1, 2, 3)
    for split in only:
        c.rays.fill(0) if hasattr(c.rays, "fill") else None
        c.ssr_trace(split=split)
        c.sync()
        same_rays = int((c.rays.raw(0) != want_rays).any(axis=-1).sum())
        same_raw = int((c.raw.raw(0) != want_raw).any(axis=-1).sum())
        parked = None
        print(f"split after {split} rounds: {timed(split):.4f} ms; texels differing: rays {same_rays}, raw {same_raw}", flush=True)


if __name__ == "__main__":
    main()
