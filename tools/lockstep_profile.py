#!/usr/bin/env python3
"""Per-rank compute time of a multi-GPU strip decomposition, measured on ONE GPU: every rank is an in-process C++ tiled
frame at the production strip size, advanced in lockstep (tiling.native_lockstep_frame: the exchanges played by copies),
with HIP-event task timing per rank.  Tells how well a decomposition balances (the multi-GPU frame time is the slowest
rank's) — everything except the wire.

    python tools/lockstep_profile.py --world 8 --frame 15360x8640 --frames 3 [--balance 2]

--balance N: after measuring equal strips, re-cut the frame N times with vkrh_balance_rows (what bench.py does before its
timed region at N > 1) and measure again.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from vk_renderer_amd import host  # noqa: E402
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.tiling import TiledFrame, native_lockstep_frame  # noqa: E402


def measure(W, H, world, bounds, frames, warmup, device):
    ranks = [TiledFrame(FrameSetup(W, H), r, world, 1, world, device, native=True, comm=None, row_bounds=bounds, force_tiled=world == 1) for r in range(world)]
    for t in ranks:
        t.prepare()
    for _ in range(warmup):
        native_lockstep_frame(ranks)
    for t in ranks:
        t.frame.enable_task_timing(True)
    torch.cuda.synchronize()
    for _ in range(frames):
        native_lockstep_frame(ranks)
    for t in ranks:
        t.flush()
    torch.cuda.synchronize()
    out = []
    for r, t in enumerate(ranks):
        times = {k: v[0] / frames for k, v in t.frame.collect_task_times().items()}
        out.append({"rank": r, "rows": t.th, "compute_ms": sum(times.values()), "per_pass_ms": times})
        t.frame.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frame", default="15360x8640")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--balance", type=int, default=0)
    args = ap.parse_args()
    W, H = (int(v) for v in args.frame.split("x"))
    world = args.world
    device = torch.device("cuda", 0)
    bounds = [r * (H // world) for r in range(world + 1)]
    passes = []
    for it in range(args.balance + 1):
        ranks = measure(W, H, world, bounds, args.frames, args.warmup, device)
        ms = [x["compute_ms"] for x in ranks]
        passes.append({"bounds": bounds, "ranks": ranks, "slowest_rank_ms": max(ms), "mean_rank_ms": sum(ms) / world})
        print(f"strips {[x['rows'] for x in ranks]}: " + " ".join(f"{v:.3f}" for v in ms) + f"  slowest {max(ms):.3f} mean {sum(ms) / world:.3f} ms",
              file=sys.stderr)
        if it < args.balance:
            bounds = host.balance_rows(ms, bounds, align=16, min_rows=max(256, H // (4 * world) // 16 * 16))
    print(json.dumps({"frame": [W, H], "world": world, "passes": passes}))


if __name__ == "__main__":
    main()
