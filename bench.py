#!/usr/bin/env python3
"""bench.py — throughput of the post-process hot path (Hi-Z + SSR + GTAO + TAA composite).

One "step" = one steady-state frame of the chain of main.cpp:347-391 on a resident synthetic
G-buffer: DownsampleGbuffer, DownsampleDepth x (L-2), SSSR_trace, SSSR_filter, SSSR_blur,
GTAO_main, GTAO_filter, GTAO_accumulate, TAA, then the history remaps of main.cpp:416-420.
N = 1 runs BASELINE.json configs[1] (3840x2160, the configuration the metric is quoted on).  N > 1
(launched by torch.distributed.run, one rank per GPU) runs BASELINE.json configs[3]: the 15360x8640
frame (16:9 camera) cut into N horizontal strips of 15360 x 8640/N, exchanging the Hi-Z pyramid /
hit-colour surfaces and history halos over RCCL every frame.  N = 2, 4, 8 cut the SAME frame
("strong" scaling between them); `value` is pixels per second, so it compares across N.

Prints ONE JSON line on rank 0.  `value` = full-resolution pixels of the whole frame processed
per second (all ranks), with every input resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import statistics
import sys
import time

# ROCr reads HSA_* once, at hsa_init: this must be in the environment before anything touches the GPU
# (the host driver only supports dmabuf IPC; without it RCCL fails with hipIpcGetMemHandle: invalid argument).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# N > 1: two frames in flight (host/frame.cpp pipelined_step; VKR_TILED_PIPELINE=0 restores the one-frame order).  The synthetic
# G-buffer is static, which is what the mode needs: the next frame's G-buffer resident when this frame's trace has run.
os.environ.setdefault("VKR_TILED_PIPELINE", "1")
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # one node: let RCCL bootstrap over the loopback interface (the container's hostname may not resolve, and by default
    # NCCL skips `lo` unless it is named); a launcher that knows better sets the variable itself
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _gpus_requested(argv):
    """--gpus N from the command line, before argparse (and before anything that could touch the GPU) runs"""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks as CHILD processes through
    torch.distributed.run — before this process has imported torch or loaded a HIP library, and never by exec — relay the
    one JSON line rank 0 prints, and leave with the children's status.  A run that does not finish within
    VKR_BENCH_LAUNCH_TIMEOUT seconds (default 1500) is killed with its whole process group and the exit status is 124."""
    import signal
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, VKR_BENCH_SELF_LAUNCHED="1")
    limit = float(os.environ.get("VKR_BENCH_LAUNCH_TIMEOUT", "1500"))
    print(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks ({' '.join(cmd[1:9])} ...)", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = child.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"[bench] the {n}-rank run did not finish within {limit:.0f} s: killing it", file=sys.stderr, flush=True)
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        child.wait()
        return 124
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.strip()]
    if child.returncode == 0 and len(lines) != 1:
        print(f"[bench] expected ONE line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    for ln in lines:
        print(ln, flush=True)
    return child.returncode


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _gpus_requested(sys.argv[1:]) > 1:
    sys.exit(self_launch(_gpus_requested(sys.argv[1:])))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from vk_renderer_amd import abi, host  # noqa: E402
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.tiling import TiledFrame, grid_for  # noqa: E402

METRIC = "Mpixels/s (GTAO+Hi-Z+SSR+TAA composite) at 4K; achieved HBM GB/s vs peak"
TILE_W, TILE_H = 3840, 2160          # N = 1: BASELINE configs[1]
C4_W, C4_H = 15360, 8640             # N > 1: BASELINE configs[3], cut into N strips
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

# Algorithmic bytes per FULL-RES pixel of each task: surface-compulsory model of SURVEY.md 8(d)
# (every input surface read once, every output written once, at storage-format size).
BYTES_PER_PX = {
    "GbufferPass": 20.0,  # --raster only: five 4-byte attachments written once (geometry / textures excluded)
    "DeferedShading": 23.0,
    "DownsampleGbuffer": 15.0,
    "DownsampleDepth": 1.667,  # the whole chain mips 2..L-1 (one task, two launches)
    "SSSR_trace": 10.333,
    "SSSR_filter": 16.0,
    "SSSR_blur": 14.0,
    "GTAO_main": 13.0,
    "GTAO_filter": 3.5,
    "GTAO_accumulate": 5.5,
    "TAA": 32.0,
}
COMPOSITE_BYTES_PER_PX = 111.0


def measured_read_bandwidth(device):
    """float4 streaming-read microbenchmark (vkr_stream_read): GB/s actually reachable on this device."""
    lib = abi.product()
    n = 2 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device)
    src.random_(0, 255)
    sink = torch.zeros(4096, dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream(device)
    best = 0.0
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        abi.check(lib.vkr_stream_read(src.data_ptr(), n, sink.data_ptr(), 4096, stream.cuda_stream), lib)
        e1.record(stream)
        e1.synchronize()
        best = max(best, n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src
    return best


def cpu_baseline(frame, setup, threads=None):
    """The oracle (CPU restatement of the reference shaders, "port") timed on this host on ONE full
    frame of the same 3840x2160 workload, from the same G-buffer bytes."""
    from oracle import binding  # the checker, used here only as the timed CPU baseline
    from vk_renderer_amd.chain import PostFxChain

    binding.install()
    ref = PostFxChain(setup.width, setup.height, backend="oracle", setup=setup)
    cores = ref.lib.vkr_ref_threads()
    for name in ("depth", "prev_depth", "normal", "albedo", "material", "velocity", "pdf", "taa_hist", "acc_hist", "blurred_hist"):
        getattr(ref, name).upload(frame.download(name).host)
    t0 = time.perf_counter()
    ref.frame()
    dt = time.perf_counter() - t0
    return {
        "value": setup.width * setup.height / dt / 1e6,
        "unit": "Mpixels/s",
        "cores": cores,
        "kind": "port",
        "sample": f"1 frame of the same {setup.width}x{setup.height} chain (CPU restatement of reference shaders, OpenMP, {dt:.1f} s)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--raster", action="store_true",
                    help="N=1 only: the G-buffer comes from the raster stage (procedural mesh scene, SceneRenderer::draw_taa) "
                         "and GbufferPass is part of every timed step")
    ap.add_argument("--rehearse-tiled", action="store_true",
                    help="N=1 only: run the multi-GPU code path (RCCL gathers, staged frame) on a one-rank group")
    ap.add_argument("--frame", type=str, default=None,
                    help=f"whole frame, WxH.  Default: {TILE_W}x{TILE_H} at N = 1 (c2), {C4_W}x{C4_H} at N > 1 (c4: N strips of W x H/N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-noskip", action="store_true", help="leave out the second timed loop with the empty-tile skips off (profiling runs: one population of launches)")
    ap.add_argument("--overlap", action="store_true",
                    help="let the rendergraph spread independent passes of a frame over several streams (measured slower: "
                         "the passes are VALU-bound, see DESIGN.md section 3)")
    ap.add_argument("--config", default=None, choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs: c1 1920x1080 GTAO main only (non-MIS); c2 3840x2160 composite (default at N = 1, the "
                         "metric's config); c3 7680x4320 composite (analytic scene: Sponza.bin is absent from the reference mount); "
                         "c4 15360x8640 composite tiled over the ranks (default at N > 1); c5 3840x2160 with 8 x (trace, filter, blur) + TAA")
    ap.add_argument("--material", default="flat", choices=["flat", "textured"],
                    help="synthetic scene: 'flat' = one roughness per object (the frozen scene the metric is quoted on); 'textured' = "
                         "roughness perturbed per texel (VKR_SYNTH_TEXTURED_ROUGHNESS): the blur's sigma varies inside every wavefront")
    ap.add_argument("--shading", action="store_true", help="add the deferred-shading composite (SURVEY 8(f) #1) between GTAO and TAA")
    args = ap.parse_args()
    if args.config is None:
        args.config = "c2" if args.gpus == 1 else "c4"
    if args.config in ("c1", "c3", "c5") and args.gpus != 1:
        raise SystemExit("--config c1/c3/c5 are single-GPU configurations")
    if args.frame is None:
        args.frame = {"c1": "1920x1080", "c2": f"{TILE_W}x{TILE_H}", "c3": "7680x4320", "c4": f"{C4_W}x{C4_H}", "c5": f"{TILE_W}x{TILE_H}"}[args.config]

    # Exactly ONE line may reach stdout (the JSON result): libraries print there too (RCCL writes its
    # version banner to stdout on rank 0), so fd 1 is pointed at stderr for the whole run and the
    # result is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # Rehearsal of the N > 1 code path on a one-GPU box (tests/test_bench_multirank_gpu.py): every rank on cuda:0 and
    # gloo instead of RCCL, which refuses two ranks on one device.  Never set by the driver.
    share_gpu = os.environ.get("VKR_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    device = torch.device("cuda", 0 if share_gpu else local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist

    # Control plane (barrier, max-reduction of the clock, handing the RCCL id around): torch.distributed over gloo.
    # Data plane: the C++ tiled frame issues grouped RCCL launches itself (vkr_all_gather / vkr_halo_exchange through the
    # communicator made here).  VKR_EXCHANGE=torch selects the older path where tiling.py issues the exchanges through
    # torch.distributed's RCCL backend; it is also the fallback when the native communicator cannot be made.
    comm, exchange = None, "none"
    if world > 1 or args.rehearse_tiled:
        if world == 1:  # rehearsal: one-rank groups, rendezvous on the loopback address
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        want_torch = os.environ.get("VKR_EXCHANGE") == "torch"
        # On a shared GPU real RCCL refuses two ranks on one device: the native wire then needs VKR_RCCL_LIBRARY to name a
        # stand-in (tests/stub_rccl: host-staged through shared memory); without one the rehearsal goes through tiling.py on gloo.
        native_possible = not share_gpu or bool(os.environ.get("VKR_RCCL_LIBRARY"))
        if want_torch and not share_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            exchange = "torch.distributed RCCL (VKR_EXCHANGE=torch)"
        elif not native_possible or want_torch:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            exchange = "torch.distributed gloo, host-staged (one-GPU rehearsal)"
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

            def share(ident):
                box = [ident]
                dist.broadcast_object_list(box, src=0)
                return box[0]

            def agree(ok):  # collective AND on the control plane: every rank takes the same branch
                t = torch.tensor([1 if ok else 0], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return int(t.item()) == 1

            why = ""
            try:
                comm = abi.Comm(rank, world, share, agree)
            except Exception as e:  # noqa: BLE001 — Comm raises on every rank or on none (see its docstring)
                why = f"{type(e).__name__}: {e}"
            if agree(comm is not None):
                # one-time self check of the wire: known bytes through all three exchanges, verified on every rank
                if not comm.self_check(device, agree):
                    why = "self check failed" + (f": {comm.self_check_error}" if comm.self_check_error else " on another rank")
                    comm.close()
                    comm = None
            elif comm is not None:
                comm.close()
                comm = None
            if comm is not None:
                exchange = "native RCCL (C++ tiled frame: vkr_all_gather / vkr_halo_exchange)"
                if os.environ.get("VKR_RCCL_LIBRARY"):
                    exchange += f" through {os.path.basename(os.environ['VKR_RCCL_LIBRARY'])}"
            else:
                print(f"[bench] native RCCL communicator unavailable ({why or 'failed on another rank'}): falling back to torch.distributed", file=sys.stderr)
                if not share_gpu:
                    dist.destroy_process_group()
                    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
                exchange = ("torch.distributed gloo, host-staged" if share_gpu else "torch.distributed RCCL") + " (fallback: " + (why or "native communicator failed on another rank") + ")"

    W, H = (int(v) for v in args.frame.lower().split("x"))
    cols, rows = grid_for(world)
    if W % cols or H % rows:
        raise SystemExit(f"frame {W}x{H} does not divide into a {cols}x{rows} grid")
    tw, th = W // cols, H // rows
    setup = FrameSetup(W, H, use_mis=0 if args.config == "c1" else 1, material=args.material)
    tiled = TiledFrame(setup, rank, world, cols, rows, device, force_tiled=args.rehearse_tiled, native=comm is not None, comm=comm)
    frame = tiled.frame
    frame.set_async(args.overlap)
    if args.config == "c1":      # GTAO main pass only (BASELINE configs[0]); non-MIS: 1+4 read, 2 written = 7 B/px
        tiled.stage_plan = [host.STAGE_GTAO_MAIN_ONLY]
        BYTES_PER_PX["GTAO_main"] = 7.0
    elif args.config == "c5":    # 8 rays per pixel = 8 x (trace, filter, blur) with frame_random = 0..7, then TAA
        tiled.stage_plan = [host.STAGE_DOWNSAMPLE] + [host.STAGE_SSR] * 8 + [host.STAGE_TAA]
    elif args.shading:
        tiled.stage_plan = [host.STAGE_CHAIN | host.STAGE_SHADING]
    if args.shading or args.config == "c5":
        frame.run(host.STAGE_BRDF_LUT)

    if args.raster:
        if world != 1 or args.rehearse_tiled:
            raise SystemExit("--raster is a single-GPU option")
        from vk_renderer_amd import scene as scn

        frame.load_scene(scn.procedural_scene(detail=96))  # 5 draws, 37k triangles
        frame.run(host.STAGE_LUT)
        frame.set_camera(setup.prev_view, setup.prev_view, setup.proj, setup.fazz)
        frame.run(host.STAGE_RASTER | host.STAGE_DOWNSAMPLE)
        frame.end_frame(swap_depth=True)  # the previous camera's depth + Hi-Z become prev_depth (main.cpp:416)
        frame.set_camera(setup.view, setup.prev_view, setup.proj, setup.fazz)
        frame.run(host.STAGE_RASTER)
        tiled._seed_histories_gpu()
        tiled.backend.sync()
        tiled.stage_plan = [host.STAGE_RASTER | (tiled.stage_plan[0] if tiled.stage_plan else host.STAGE_CHAIN)]
        tiled.prepare = lambda: None
    tiled.prepare()  # LUT, G-buffer (tile + halo), prev depth, histories
    if args.config == "c1":
        frame.run(host.STAGE_DOWNSAMPLE)  # GTAO reads depth mip 1; built once, outside the timed pass
    for _ in range(2):  # first touches (allocator pools, lazily created pipelines); the W warm-up steps proper run right before the timed region
        tiled.step()
    tiled.flush()

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Strips balanced by cost (N > 1, C++ tiled frame): the rows of a frame do not cost the same — a strip of sky is cheap, a
    # strip of reflecting floor is not (tools/lockstep_profile.py: 1.15 .. 2.11 ms for the eight equal strips of c4) — and
    # the frame takes as long as the slowest rank.  Before the timed region every rank measures its passes (events around
    # the kernels: no waiting for exchanges in it), the times are shared, vkrh_balance_rows cuts the frame where the
    # cumulative cost is r / N of the total, and the frame is rebuilt on the new strips (their shares then travel with
    # vkr_all_gather_v).  Twice: the cost inside a strip is not uniform either.  VKR_BALANCE=0 keeps equal strips.
    row_bounds = [r * th for r in range(world + 1)]
    balance_log = []
    # (VKR_BALANCE_REBUILD=1, tests: the one-rank rehearsal goes through the same measure / share / rebuild sequence although
    # its single strip cannot move)
    force_rebuild = os.environ.get("VKR_BALANCE_REBUILD") == "1"
    any_frame = os.environ.get("VKR_BALANCE_ANY_FRAME") == "1"  # tests: re-cut the strips of a small rehearsal frame too
    if ((world > 1 and (args.config == "c4" or any_frame)) or (args.rehearse_tiled and force_rebuild)) and comm is not None and os.environ.get("VKR_BALANCE", "1") != "0":
        for _ in range(2):
            BAL_STEPS = 3
            frame.enable_task_timing(True)
            for _ in range(BAL_STEPS):
                tiled.step()
            tiled.flush()
            barrier()
            mine = sum(v[0] for v in frame.collect_task_times().values()) / BAL_STEPS
            frame.enable_task_timing(False)
            every = [None] * world
            dist.all_gather_object(every, float(mine))
            try:
                new_bounds = host.balance_rows(every, row_bounds, align=16, min_rows=max(256, H // (4 * world) // 16 * 16))
            except RuntimeError as e:  # e.g. a --frame whose height is not a multiple of 16: the same on every rank, keep the strips
                print(f"[bench] strips not re-cut: {e}", file=sys.stderr)
                new_bounds = row_bounds
            balance_log.append({"rows": [row_bounds[r + 1] - row_bounds[r] for r in range(world)], "compute_ms": [round(v, 4) for v in every]})
            if new_bounds == row_bounds and not force_rebuild:
                break
            row_bounds = new_bounds
            frame.close()
            tiled = TiledFrame(setup, rank, world, cols, rows, device, force_tiled=args.rehearse_tiled, native=True, comm=comm, row_bounds=row_bounds)
            frame = tiled.frame
            frame.set_async(args.overlap)
            tiled.prepare()
            for _ in range(max(args.warmup, 2)):
                tiled.step()
            tiled.flush()
            barrier()

    # Per-pass times come from a short calibration run with an event pair around every pass; the timed region then
    # keeps only the pair around the dominant pass (the roofline's live measurement): an event record costs ~3.5 us
    # of queue time, and nine pairs per frame would slow the 1 ms frame that is being measured by 7 %.
    CAL_STEPS = 10
    frame.enable_task_timing(True)
    timed_waits = tiled.native and frame.tiled_handle is not None and comm is not None
    if timed_waits:  # how long the compute stream stands still for each exchange (diagnostics of the multi-GPU wire)
        frame.tiled_time_waits(True)
    for _ in range(CAL_STEPS):
        tiled.step()
    tiled.flush()
    barrier()
    exchange_wait_ms = None
    if timed_waits:
        mine = {k: v / CAL_STEPS for k, v in frame.tiled_wait_times().items()}
        frame.tiled_time_waits(False)
        every = [None] * world
        if world > 1:
            dist.all_gather_object(every, mine)
        else:
            every = [mine]
        exchange_wait_ms = {k: [round(e[k], 4) for e in every] for k in mine}  # per rank
    # bytes every rank RECEIVES per frame over the wire: the other ranks' shares of the gathered surfaces + its halo rows
    exchange_bytes_per_rank = None
    if tiled.native and frame.tiled_handle is not None and world > 1:
        by_gather = frame.albedo_by_gather
        mine = {"hiz": sum(p[2] for p in frame.tiled_gather_parts(0)), "albedo": sum(p[2] for p in frame.tiled_gather_parts(1)) if by_gather else 0,
                "halo": sum(p[3] for s_ in range(3) for p in frame.tiled_halo_peers(s_)), "hit": 0 if by_gather else frame.tiled_hit_bytes()}
        every = [None] * world
        dist.all_gather_object(every, mine)
        # hit colours: by request / reply (4-byte requests in + 16-byte replies in, last frame's segments) or the all-gathered albedo
        exchange_bytes_per_rank = [{"hiz_gather_in": sum(e["hiz"] for e in every) - every[r]["hiz"],
                                    "hit_colours_in": (sum(e["albedo"] for e in every) - every[r]["albedo"]) if by_gather else every[r]["hit"],
                                    "hit_colours": "all-gather of the albedo" if by_gather else "request / reply",
                                    "halo_in": every[r]["halo"]} for r in range(world)]
        for e in exchange_bytes_per_rank:
            e["total_in"] = e["hiz_gather_in"] + e["hit_colours_in"] + e["halo_in"]
    calibration = frame.collect_task_times()
    per_pass_ms = {k: v[0] / CAL_STEPS for k, v in calibration.items()}          # all executions of the task in one step
    launches_per_step = {k: v[1] / CAL_STEPS for k, v in calibration.items()}    # executions of the task per step (c5: 8 x SSR)
    dominant = max(per_pass_ms, key=per_pass_ms.get)
    frame.enable_task_timing(True, only=dominant)
    # one event per step boundary on the frame's stream (= torch's current stream): the median step time of SURVEY 8(d)
    stream = torch.cuda.current_stream(device)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if args.steps >= 10 else []
    # The W untimed warm-up steps, immediately ahead of the timed region: everything above (balancing, the calibration run and
    # the host-side reading of its events) leaves the device idle for a moment, and the first frames after an idle gap run at a
    # lower clock (measured: a 20-step loop right after such a gap 0.682 ms per frame, the same loop in steady state 0.650).
    for _ in range(args.warmup):
        tiled.step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if marks:
            marks[i].record(stream)
        tiled.step()
    if marks:
        marks[-1].record(stream)
    tiled.flush()  # the halo refreshes of the last frame are still in flight: they belong to the timed work
    barrier()
    elapsed = time.perf_counter() - t0
    task_times = frame.collect_task_times()
    frame.enable_task_timing(False)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)] if marks else []

    # The headline depends on the frame's content: tiles of the blur / filter without a reflection / hit skip their taps
    # (bit-identical output).  A second, short timed loop with the skips switched off gives the content-independent number.
    noskip_ms = generic_ms = None
    if world == 1 and args.config in ("c2", "c3", "c4", "c5") and not args.no_noskip:
        lib = abi.product()
        before = lib.vkr_get_switches()
        NOSKIP_STEPS = max(10, min(args.steps, 20))

        def timed_with(switches):
            lib.vkr_set_switches(before | switches)
            for _ in range(max(10, args.warmup)):  # (as many warm-up frames as the main loop had: see there)
                tiled.step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(NOSKIP_STEPS):
                tiled.step()
            tiled.flush()
            barrier()
            dt = (time.perf_counter() - t1) / NOSKIP_STEPS * 1e3
            lib.vkr_set_switches(before)
            return dt

        noskip_ms = timed_with(abi.SWITCH_BLUR_NO_SKIP | abi.SWITCH_FILTER_NO_SKIP)
        # ... and with the blur's wave-uniform-sigma path off as well: nothing left that depends on what the frame shows
        generic_ms = timed_with(abi.SWITCH_BLUR_NO_SKIP | abi.SWITCH_FILTER_NO_SKIP | abi.SWITCH_BLUR_GENERIC)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        px = W * H
        tile_px = tiled.window[2] * tiled.window[3]  # pixels this rank's kernels actually process (tile + halo)
        # One execution of a task processes the whole window once (the two launches of the Hi-Z chain are ONE
        # DownsampleDepth task; the eight SSSR_blur executions of c5 each move the full bytes).
        avg_launch_ms = task_times[dominant][0] / task_times[dominant][1]  # live, over the timed region
        algo_bytes_launch = BYTES_PER_PX[dominant] * tile_px
        achieved = algo_bytes_launch / (avg_launch_ms * 1e-3) / 1e9
        measured_read = measured_read_bandwidth(device)

        def from_profiles(fname):
            path = os.path.join(ROOT, "profiles", fname)
            if not os.path.exists(path):
                return None, None
            with open(path) as f:
                d = json.load(f).get(args.config)  # one digest per BASELINE config
            if not d:
                return None, None
            note = ""
            if d.get("_library_sha16"):  # was the digest taken with the kernel library this run has loaded?
                import hashlib

                with open(abi.PRODUCT_LIB, "rb") as lf:
                    mine = hashlib.sha256(lf.read()).hexdigest()[:16]
                note = "; SAME kernel library as this run" if mine == d["_library_sha16"] else f"; digest of ANOTHER build of the kernel library ({d['_library_sha16']}, this run {mine})"
            return d.get(dominant), f"profiles/{fname}[{args.config}]" + (f" ({d['_source']}{note})" if "_source" in d else "")

        # HBM bytes / VALU-busy of the dominant kernel are NOT measured in this run: they are the committed rocprofv3
        # --pmc digests of the same command (tools/profile_run.sh + tools/profile_digest.py --config), kept per BASELINE
        # config and only valid for that config's own frame, the frozen scene and one GPU.
        own_frame = world == 1 and args.material == "flat" and not args.shading and not args.raster and (W, H) == {
            "c1": (1920, 1080), "c2": (TILE_W, TILE_H), "c3": (7680, 4320), "c4": (C4_W, C4_H), "c5": (TILE_W, TILE_H)}[args.config]
        traffic, traffic_src = from_profiles("traffic.json") if own_frame else (None, None)
        valu_busy, valu_src = from_profiles("valu_busy.json") if own_frame else (None, None)
        workload = {"c1": f"{W}x{H} synthetic G-buffer: GTAO main pass only (non-MIS)",
                    "c5": f"{W}x{H} synthetic G-buffer: Hi-Z downsample + 8 x SSR (trace, filter, blur) + TAA"}.get(
            args.config, f"{W}x{H} synthetic G-buffer: Hi-Z downsample + SSR (trace, filter, blur) + GTAO (main, filter, accumulate)"
                         + (" + deferred shading" if args.shading else "") + " + TAA"
                         + (f", tiled over {world} GPUs as {cols}x{rows} strips with RCCL halo / Hi-Z / hit-colour exchange" if world > 1 else ""))
        out = {
            "metric": METRIC,
            "value": px * args.steps / elapsed / 1e6,
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            # N = 2, 4, 8 cut the same 15360x8640 frame (total work fixed); N = 1 is the 3840x2160 frame the metric is quoted on
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "baseline_config": args.config if (W, H) == {"c1": (1920, 1080), "c2": (TILE_W, TILE_H), "c3": (7680, 4320),
                                                             "c4": (C4_W, C4_H), "c5": (TILE_W, TILE_H)}[args.config] else None,
                "frame": [W, H],
                "tile_per_gpu": [tw, th],
                "grid": [cols, rows],
                "halo_px": tiled.halo,
                "gathered_hiz_mips": tiled.gather_mips if tiled.tiled else 0,
                "strip_rows": [row_bounds[r + 1] - row_bounds[r] for r in range(world)],
                "strip_balance": balance_log,  # per balancing pass: the strips and every rank's compute time with them
                "exchange": exchange,
                "launcher": "self (bench.py started its ranks)" if os.environ.get("VKR_BENCH_SELF_LAUNCHED") == "1" else ("torch.distributed.run" if world > 1 else "none"),
                "gbuffer": "rasterised procedural mesh scene (GbufferPass timed)" if args.raster else "analytic generator (not timed)",
                "material": args.material + (" (one roughness per object)" if args.material == "flat" else " (roughness per texel, VKR_SYNTH_TEXTURED_ROUGHNESS)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "frac_of_measured": achieved / measured_read,  # against the float4 stream-read rate measured in this run
                "traffic": traffic,
                "traffic_source": traffic_src,   # committed PMC digest, not measured in this run
                # fraction of the SIMD issue slots the kernel keeps busy with VALU work (SQ_ACTIVE_INST_VALU):
                # why the HBM fraction is low — the pass is issue-bound, not memory-bound
                "valu_busy": valu_busy,
                "valu_busy_source": valu_src,
                "avg_launch_ms": avg_launch_ms,
                "launches_timed": task_times[dominant][1],
                "algorithmic_bytes_per_launch": algo_bytes_launch,
            },
            "composite_gbps": sum(BYTES_PER_PX.get(k, 0.0) * n for k, n in launches_per_step.items()) * tile_px / (elapsed / args.steps) / 1e9 * world,
            # calibration run (CAL_STEPS steps with an event pair around every pass) just before the timed region, NOT the timed region itself
            "per_pass_ms": per_pass_ms,
            "per_pass_source": f"calibration run of {CAL_STEPS} steps before the timed region (event pair around every pass)",
            "per_pass_gbps": {k: BYTES_PER_PX[k] * tile_px * launches_per_step[k] / (v * 1e-3) / 1e9
                              for k, v in per_pass_ms.items() if k in BYTES_PER_PX and v > 0},
            "exchange_ms": tiled.exchange_ms(args.steps),
            # calibration run: ms per frame every rank's compute stream stood still for each exchange (None: no wire)
            "exchange_wait_ms": exchange_wait_ms,
            "exchange_bytes_per_rank": exchange_bytes_per_rank,
            # native wire, rank 0: hit-colour rounds (enqueued on the previous frame's capacities, exact after a host round trip, repeated after an overflow)
            "hit_rounds": list(frame.tiled_hit_rounds()) if (tiled.native and frame.tiled_handle is not None and comm is not None and not frame.albedo_by_gather) else None,
            "tiled_options": {"gather_mode": frame.gather_mode, "all_gather_v": "broadcast" if os.environ.get("VKR_GATHER_V_BROADCAST") == "1" else "point-to-point",
                              "trace_local_rows_first": frame.tiled_local_first(),
                              "rows_computed": "whole window" if os.environ.get("VKR_TILED_WHOLE_WINDOW", "0") not in ("", "0") else "the rows that are read",
                              # 2: the next frame's downsample and depth all-gather start right after this frame's trace (host/frame.cpp:
                              # pipelined_step; the benchmark's G-buffer is static, so the next frame's is resident); every step runs every pass once
                              "frames_in_flight": 2 if frame.tiled_pipelined() else 1} if (tiled.native and frame.tiled_handle is not None) else None,
            "measured_read_gbps": measured_read,
        }
        if noskip_ms is not None:
            out["ms_per_step_noskip"] = noskip_ms
            out["value_noskip"] = px / (noskip_ms * 1e-3) / 1e6
            out["noskip_note"] = "second timed loop with VKR_SWITCH_BLUR_NO_SKIP | VKR_SWITCH_FILTER_NO_SKIP: every tap evaluated, same output"
        if generic_ms is not None:
            out["ms_per_step_generic"] = generic_ms
            out["value_generic"] = px / (generic_ms * 1e-3) / 1e6
            out["generic_note"] = ("third timed loop: the skips off AND VKR_SWITCH_BLUR_GENERIC (every blur wave on the per-lane-sigma loop): "
                                   "the time of a frame whose content helps nowhere")
        if world > 1 and (W, H) == (C4_W, C4_H):
            # the denominator of the scaling curve: the SAME 15360x8640 frame on one GPU (python bench.py --config c4 --gpus 1)
            # (a committed measurement of another run, another box and possibly another build: its git head travels with it, and
            # the driver's own N = 1 ... 8 runs of one invocation are the scaling curve that counts)
            ref_path = os.path.join(ROOT, "profiles", "r04_bench_c4_n1.json")
            if os.path.exists(ref_path):
                with open(ref_path) as f:
                    ref = json.load(f)
                out["single_gpu_same_frame_ms"] = ref["ms_per_step"]
                out["single_gpu_same_frame_source"] = ("profiles/r04_bench_c4_n1.json (python bench.py --config c4 --gpus 1 on one MI355X, build "
                                                       + str(ref.get("git_head", "?"))[:12] + ")")
                out["speedup_vs_single_gpu_same_frame"] = ref["ms_per_step"] / (elapsed / args.steps * 1e3)
        if step_ms:  # median of the hipEvent-timed frames (>= 10 steps), beside the contract's wall-clock mean
            med = statistics.median(step_ms)
            out["ms_per_step_median"] = med
            out["value_median"] = px / (med * 1e-3) / 1e6  # rank 0's stream only: meaningful at N = 1
        if world == 1 and not args.no_cpu_baseline and args.config == "c2" and not args.shading and not args.rehearse_tiled:
            out["cpu_baseline"] = cpu_baseline(frame, setup)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if comm is not None:
        torch.cuda.synchronize(device)
        frame.close()
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
