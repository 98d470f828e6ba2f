#!/usr/bin/env python3
"""Per-rank compute time of a multi-GPU grid, measured on ONE GPU: every rank of the grid is an in-process
TiledFrame at the production tile size, advanced in lockstep (tiling.TiledFrame.phases(), the exchanges played
by copies), with HIP-event task timing per rank.  Tells how well a decomposition balances (the multi-GPU frame
time is the slowest rank's) and what the pack / scatter launches cost — everything except the wire.

    python tools/lockstep_profile.py --grid 4x2 --tile 3840x2160 --frames 5
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.tiling import TiledFrame  # noqa: E402


def move_halos(ranks, which):
    for r, t in enumerate(ranks):
        for nb, _, rbuf in t.halo_peers(which):
            if rbuf is not None:
                rbuf.copy_([p for p in ranks[nb].halo_peers(which) if p[0] == r][0][1])


def lockstep_frame(ranks):
    world = len(ranks)
    gens = [t.phases() for t in ranks]
    while True:
        ops = [next(g, None) for g in gens]
        if ops[0] is None:
            return
        kind = ops[0][0]
        if kind == "gather_wait":
            for _, g in ops:
                for i, (_, recv) in enumerate(g.parts):
                    for src in range(world):
                        recv.view(world, -1)[src].copy_(ops[src][1].parts[i][0])
        elif kind == "halo_wait":
            move_halos(ranks, ops[0][1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="4x2")
    ap.add_argument("--tile", default="3840x2160")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    cols, rows = (int(v) for v in args.grid.split("x"))
    tw, th = (int(v) for v in args.tile.split("x"))
    world = cols * rows
    device = torch.device("cuda", 0)
    W, H = tw * cols, th * rows
    ranks = [TiledFrame(FrameSetup(W, H), r, world, cols, rows, device, force_tiled=True) for r in range(world)]
    for t in ranks:
        t.prepare()
    for _ in range(args.warmup):
        lockstep_frame(ranks)
    for t in ranks:
        t.frame.enable_task_timing(True)
    torch.cuda.synchronize()
    for _ in range(args.frames):
        lockstep_frame(ranks)
    torch.cuda.synchronize()
    out = {"grid": [cols, rows], "frame": [W, H], "tile": [tw, th], "ranks": []}
    for r, t in enumerate(ranks):
        times = {k: v[0] / args.frames for k, v in t.frame.collect_task_times().items()}
        out["ranks"].append({"rank": r, "window": list(t.window), "compute_ms": sum(times.values()), "per_pass_ms": times})
    worst = max(x["compute_ms"] for x in out["ranks"])
    mean = sum(x["compute_ms"] for x in out["ranks"]) / world
    out["slowest_rank_ms"], out["mean_rank_ms"] = worst, mean
    print(json.dumps(out))
    for x in out["ranks"]:
        print(f"rank {x['rank']} window {x['window']}: {x['compute_ms']:.3f} ms  " +
              " ".join(f"{k}={v:.3f}" for k, v in x["per_pass_ms"].items()), file=sys.stderr)
    print(f"slowest {worst:.3f} ms, mean {mean:.3f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
