#!/usr/bin/env python3
"""The multi-GPU frame's schedule against realistic wire times, on ONE GPU (measurement; DESIGN_MULTIGPU.md "Emulated wire").

All N ranks of the strip decomposition live in this process as C++ tiled frames.  First the in-process harness drives them in
lockstep with real data (tiling.native_lockstep_frame: the wire played by copies), the last frame with the hit segments laid out
as the native frame lays them out (host.hit_capacities).  The benchmark scene is static, so from then on every exchange of every
frame delivers exactly the bytes its receive buffers already hold.  Then, one rank at a time, the frame goes on NATIVELY
(vkrh_tiled_step: its own exchange stream, events, the hit round enqueued without the host) on an emulated communicator
(vkr_comm_create_emulated): every exchange holds the exchange stream for launch_us + (bytes on the busiest link) / link rate and
moves nothing.  Measured: ms per frame of each rank with everything the schedule overlaps or fails to overlap, and how long its
compute stream stood still for each exchange.  The frame of the N-GPU job takes the slowest rank's time.

    python tools/wire_emulation.py --world 8 --frame 15360x8640 --link-gbps 60 45 [--bounds-from profiles/r04_final_strip_balance_c4.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from vk_renderer_amd import abi  # noqa: E402
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.tiling import TiledFrame, native_lockstep_frame  # noqa: E402


def prepared_ranks(W, H, world, bounds, device, warm):
    ranks = [TiledFrame(FrameSetup(W, H), r, world, 1, world, device, native=True, comm=None, row_bounds=bounds) for r in range(world)]
    for t in ranks:
        t.prepare()
    # the SSR frame counter (frame_random, advanced_ssr.cpp:168-171) is pinned before every frame: every frame then traces the same
    # rays and asks for the same hit texels, so that what the receive buffers hold IS what the peers would send again
    for _ in range(warm):
        pin(ranks)
        native_lockstep_frame(ranks)
    pin(ranks)
    native_lockstep_frame(ranks, hit_in_capacities=True)  # the layout the native frame's next round expects in its receive buffers
    torch.cuda.synchronize()
    return ranks


def pin(ranks):
    for t in ranks:
        t.frame.pin_randoms(0.0, 0, 0)


def steps_of(t, n):
    for _ in range(n):
        t.frame.pin_randoms(0.0, 0, 0)
        t.frame.tiled_step()
    t.frame.tiled_flush()


def measure(W, H, world, bounds, device, gbps, launch_us, steps, warm):
    ranks = prepared_ranks(W, H, world, bounds, device, warm)
    counts = ranks[0].hit_matrix
    out = []
    for r, t in enumerate(ranks):
        comm = abi.Comm.emulated(r, world, gbps, launch_us)
        t.frame.tiled_emulate_wire(comm.handle, counts)
        steps_of(t, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps_of(t, steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        t.frame.tiled_time_waits(True)
        steps_of(t, steps)
        waits = {k: v / steps for k, v in t.frame.tiled_wait_times().items()}
        t.frame.tiled_time_waits(False)
        rounds = list(t.frame.tiled_hit_rounds())
        errors = t.frame.tiled_hit_errors()
        out.append({"rank": r, "rows": t.th, "ms_per_frame": ms, "exposed_wait_ms": waits, "hit_rounds": rounds, "hit_errors": errors})
        print(f"  rank {r} ({t.th} rows): {ms:.3f} ms per frame, exposed " + " ".join(f"{k} {v:.3f}" for k, v in waits.items()) + f", rounds {rounds}", file=sys.stderr)
        t.frame.close()
        comm.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frame", default="15360x8640")
    ap.add_argument("--link-gbps", type=float, nargs="*", default=[60.0, 45.0])
    ap.add_argument("--launch-us", type=float, default=15.0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warm", type=int, default=2)
    ap.add_argument("--bounds-from", default=None, help="a tools/lockstep_profile.py JSON: the strips of its last balancing pass")
    ap.add_argument("--one-gpu-ms", type=float, default=None, help="ms per frame of the same frame on one GPU (for the speed-up column)")
    args = ap.parse_args()
    W, H = (int(v) for v in args.frame.split("x"))
    world = args.world
    bounds = [r * (H // world) for r in range(world + 1)]
    if args.bounds_from:
        with open(args.bounds_from) as f:
            bounds = json.load(f)["passes"][-1]["bounds"]
        assert len(bounds) == world + 1
    device = torch.device("cuda", 0)
    result = {"frame": [W, H], "world": world, "bounds": bounds, "launch_us": args.launch_us, "runs": []}
    for gbps in args.link_gbps:
        print(f"link {gbps:g} GB/s per direction:", file=sys.stderr)
        ranks = measure(W, H, world, bounds, device, gbps, args.launch_us, args.steps, args.warm)
        slowest = max(x["ms_per_frame"] for x in ranks)
        run = {"link_gbps": gbps, "ranks": ranks, "frame_ms": slowest}
        if args.one_gpu_ms:
            run["speedup_vs_one_gpu"] = args.one_gpu_ms / slowest
        print(f"  frame {slowest:.3f} ms" + (f" = {args.one_gpu_ms / slowest:.2f} x" if args.one_gpu_ms else ""), file=sys.stderr)
        result["runs"].append(run)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
