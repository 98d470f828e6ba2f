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

(one child process per rank; the parent never touches the GPU)
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


def prepared_ranks(W, H, world, bounds, device, warm, first=None):
    # (`first`: the rank whose frame — images, streams, events — is made before the others')
    order = list(range(world)) if first is None else [first] + [r for r in range(world) if r != first]
    made = {r: TiledFrame(FrameSetup(W, H), r, world, 1, world, device, native=True, comm=None, row_bounds=bounds) for r in order}
    ranks = [made[r] for r in range(world)]
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


def measure_rank(W, H, world, bounds, device, r, rates, launch_us, steps, warm, per_frame=False):
    """rank r of the decomposition, its frame made first in this process (with eight frames in one process the ones made
    fourth and fifth ran every small kernel ten times slower — an artefact of that set-up, not of the rank: made first they
    are as fast as the others; on a node every process holds one rank)"""
    ranks = prepared_ranks(W, H, world, bounds, device, warm, r)
    counts = ranks[0].hit_matrix
    for q, t in enumerate(ranks):
        if q != r:
            t.frame.close()
    t = ranks[r]
    out = []
    for gbps in rates:
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
        tasks = None
        if per_frame:  # what the nine passes themselves take on this schedule (HIP events around every task)
            t.frame.enable_task_timing(True)
            steps_of(t, steps)
            torch.cuda.synchronize()
            tasks = {k: v[0] / steps for k, v in t.frame.collect_task_times().items()}
            t.frame.enable_task_timing(False)
        out.append({"rank": r, "rows": t.th, "link_gbps": gbps, "pipelined": t.frame.tiled_pipelined(), "ms_per_frame": ms, "exposed_wait_ms": waits, "hit_rounds": list(t.frame.tiled_hit_rounds()),
                    "hit_errors": t.frame.tiled_hit_errors(), "task_ms": tasks})
        print(f"  rank {r} ({t.th} rows) at {gbps:g} GB/s: {ms:.3f} ms per frame, exposed " + " ".join(f"{k} {v:.3f}" for k, v in waits.items()), file=sys.stderr)
    t.frame.close()
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
    ap.add_argument("--rebalance", type=int, default=0, help="re-cut the strips this many times by the ranks' frame times at the first link rate")
    ap.add_argument("--rank", type=int, default=None, help="(child) measure this rank and print its JSON")
    ap.add_argument("--task-times", action="store_true", help="also the per-pass times on the native schedule")
    args = ap.parse_args()
    W, H = (int(v) for v in args.frame.split("x"))
    world = args.world
    bounds = [r * (H // world) for r in range(world + 1)]
    if args.bounds_from:
        with open(args.bounds_from) as f:
            bounds = json.load(f)["passes"][-1]["bounds"]
        assert len(bounds) == world + 1
    if args.rank is not None:  # child: one rank, every link rate
        device = torch.device("cuda", 0)
        print(json.dumps(measure_rank(W, H, world, bounds, device, args.rank, args.link_gbps, args.launch_us, args.steps, args.warm, args.task_times)))
        return
    # parent: one child process per rank (this process never touches the GPU)
    import subprocess
    import tempfile

    def run_all(bounds):
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump({"passes": [{"bounds": bounds}]}, f)
        argv = [a for a in sys.argv[1:]]
        if "--bounds-from" in argv:
            i = argv.index("--bounds-from")
            del argv[i:i + 2]
        per_rank = []
        for r in range(world):
            cmd = [sys.executable, os.path.abspath(__file__), "--rank", str(r), "--bounds-from", f.name] + argv
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            sys.stderr.write("".join(l for l in p.stderr.splitlines(True) if l.startswith("  rank")))
            if p.returncode != 0:
                sys.stderr.write(p.stderr[-3000:])
                raise SystemExit(f"rank {r}: the child exited {p.returncode}")
            per_rank.append(json.loads(p.stdout.strip().splitlines()[-1]))
        os.unlink(f.name)
        return per_rank

    per_rank = run_all(bounds)
    # --rebalance K: re-cut the strips K times by what a rank's FRAME takes at the first link rate (its passes and what it waits
    # for), not by its passes alone — the balance a job on real links would find if it fed its per-rank frame times back
    for it in range(args.rebalance):
        from vk_renderer_amd import host

        ms = [x[0]["ms_per_frame"] for x in per_rank]
        new = host.balance_rows(ms, bounds, align=16, min_rows=max(256, H // (4 * world) // 16 * 16))
        print(f"re-cut by frame time at {args.link_gbps[0]:g} GB/s: {[new[r + 1] - new[r] for r in range(world)]} (slowest was {max(ms):.3f} ms)", file=sys.stderr)
        if new == bounds:
            break
        bounds = new
        per_rank = run_all(bounds)
    result = {"frame": [W, H], "world": world, "bounds": bounds, "launch_us": args.launch_us, "runs": []}
    for i, gbps in enumerate(args.link_gbps):
        ranks = [x[i] for x in per_rank]
        slowest = max(x["ms_per_frame"] for x in ranks)
        run = {"link_gbps": gbps, "ranks": ranks, "frame_ms": slowest, "mean_rank_ms": sum(x["ms_per_frame"] for x in ranks) / world}
        if args.one_gpu_ms:
            run["speedup_vs_one_gpu"] = args.one_gpu_ms / slowest
        print(f"link {gbps:g} GB/s: frame {slowest:.3f} ms (slowest rank), mean rank {run['mean_rank_ms']:.3f}" + (f" = {args.one_gpu_ms / slowest:.2f} x" if args.one_gpu_ms else ""), file=sys.stderr)
        result["runs"].append(run)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
