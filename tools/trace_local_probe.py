#!/usr/bin/env python3
"""GPU probe of the multi-GPU trace on ONE strip of a frame: vkr_sssr_trace_windowed (one launch, whole-frame pyramid) against
vkr_sssr_trace_windowed_head + _resume (head on the strip's own pyramid rows), with the share of rays the head parks.

    python tools/trace_local_probe.py [W H world rank]      (default: 15360 8640 8 5 — a floor strip of BASELINE config 4)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.chain import PostFxChain  # noqa: E402

HALO = 48


def main():
    a = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else [15360, 8640, 8, 5]
    W, H, world, rank = a
    th = H // world
    y0, y1 = rank * th, (rank + 1) * th
    wy0, wy1 = max(0, y0 - HALO), min(H, y1 + HALO)
    setup = FrameSetup(W, H)
    plain = PostFxChain(W, H, backend="product", device="cuda", setup=setup)
    plain.synth(); plain.downsample()
    c = PostFxChain(W, H, backend="product", device="cuda", setup=setup, window=(0, wy0, W, wy1 - wy0), force_tiled=True)
    c.synth(); c.build_prev_hiz(); c.init_histories(); c.preintegrate_pdf(); c.downsample()
    # the gathered pyramid: whole-frame image mips 1..L-1, copied on the device
    for m in range(c.frame_hiz.mips):
        src, dst = plain.depth, c.frame_hiz
        h = max(1, (H // 2) >> m)
        if m + 1 >= src.mips:
            break
        assert src.pitch[m + 1] == dst.pitch[m]
        n = dst.pitch[m] * h
        dst.tensor[dst.offset[m]: dst.offset[m] + n].copy_(src.tensor[src.offset[m + 1]: src.offset[m + 1] + n])
    for src, dst in ((c.albedo, c.frame_albedo), (c.dn, c.frame_normals)):
        o = src.origin[1] * dst.pitch[0]
        dst.tensor[o: o + src.height * src.pitch[0]].copy_(src.tensor[: src.height * src.pitch[0]])
    del plain
    torch.cuda.empty_cache()

    def timed(fn, n=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n

    total = c.rays.width * c.rays.height
    print(f"{W}x{H}, strip {rank} of {world}: window rows [{wy0}, {wy1}), {total} rays")
    print(f"one launch (whole-frame pyramid): {timed(lambda: c.ssr_trace_windowed(frame_random=0)):.4f} ms")
    c.sync()
    want = c.rays.raw(0).copy()
    for levels in (4, 3):
        for park_after in (2, 4):
            t_head = timed(lambda: c.ssr_trace_windowed_head(levels, frame_random=0, park_after=park_after))
            c.sync()
            parked = int(c._trace_workspace[:4].view(torch.int32)[0].item())
            t_both = timed(lambda: (c.ssr_trace_windowed_head(levels, frame_random=0, park_after=park_after), c.ssr_trace_windowed_resume(frame_random=0)))
            c.sync()
            same = int((c.rays.raw(0) != want).any(axis=-1).sum())
            print(f"local levels {levels}, park after {park_after} rounds: head {t_head:.4f} ms, head + resume {t_both:.4f} ms, parked {parked / total:.3f} of the rays, differing texels {same}", flush=True)


if __name__ == "__main__":
    main()
