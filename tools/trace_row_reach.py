#!/usr/bin/env python3
"""How far do the SSR marches of a strip reach vertically?  From the ORACLE (test infrastructure, never the product): per ray the
rows of pyramid level 0 covered by every texel its march fetched (oracle/passes_ssr.cpp: vkr_ref_set_reach_sink), then, for a frame
cut into N strips with the tiled frame's geometry (window = strip + 48 px, rays computed on strip + 14 half-res rows), the share of
a rank's rays whose whole march stays on texels that lie entirely inside its window — the rays a head launch could finish before the
depth all-gather has arrived if the rank built every coarse level it can from its own rows (DESIGN_MULTIGPU.md, "Local rows first").

    python tools/trace_row_reach.py --size 15360 8640 --ranks 8 4 2
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from oracle import binding  # noqa: E402
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.chain import PostFxChain  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=(3840, 2160))
    ap.add_argument("--ranks", type=int, nargs="*", default=[8])
    ap.add_argument("--halo", type=int, default=48)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trace_row_reach.json"))
    a = ap.parse_args()
    lib = binding.install()
    W, H = a.size
    c = PostFxChain(W, H, backend="oracle", setup=FrameSetup(W, H))
    c.synth(); c.build_prev_hiz(); c.init_histories(); c.preintegrate_pdf()
    c.downsample()
    h2, w2 = H // 2, W // 2
    reach = np.zeros((h2, w2, 2), dtype=np.uint16)
    lib.vkr_ref_set_reach_sink.argtypes = [C.c_void_p, C.c_int]
    lib.vkr_ref_set_reach_sink(reach.ctypes.data, reach.strides[0])
    c.ssr_trace(frame_random=0)
    lib.vkr_ref_set_reach_sink(None, 0)
    lo, hi = reach[..., 0].astype(np.int64), reach[..., 1].astype(np.int64)
    nothing = hi == 0  # fetched nothing inside the frame
    rows = np.arange(h2)[:, None]
    up, down = np.where(nothing, 0, rows - lo), np.where(nothing, 0, hi - 1 - rows)
    print(f"{W}x{H}: rays {h2 * w2}, reach above the pixel (level-0 rows): mean {up.mean():.1f} median {np.median(up):.0f} 90 % {np.percentile(up, 90):.0f} max {up.max()}; "
          f"below: mean {down.mean():.1f} median {np.median(down):.0f} 90 % {np.percentile(down, 90):.0f} max {down.max()}")
    out = {"frame": [W, H], "ranks": {}}
    for n in a.ranks:
        per = []
        for r in range(n):
            y0, y1 = r * (H // n), (r + 1) * (H // n)
            w0, w1 = max(0, y0 - a.halo) // 2, min(H, y1 + a.halo) // 2  # window rows, level 0 of the pyramid
            c0, c1 = max(w0, y0 // 2 - 14), min(w1, y1 // 2 + 14)         # rows whose rays the rank computes
            L, Hh, none = lo[c0:c1], hi[c0:c1], nothing[c0:c1]
            local = none | ((L >= w0) & (Hh <= w1))
            per.append({"rank": r, "rays": int(local.size), "stay_in_window": float(local.mean())})
        tot = sum(p["rays"] * p["stay_in_window"] for p in per) / sum(p["rays"] for p in per)
        out["ranks"][str(n)] = {"per_rank": per, "stay_in_window": tot}
        print(f"N = {n}: rays whose whole march lies on texels inside the rank's window: {tot:.3f}  per rank " + " ".join(f"{p['stay_in_window']:.2f}" for p in per))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        import json
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
