"""Frame tiling across the GPUs of one node.

The frame is cut into cols x rows tiles, one per rank.  Every rank holds its tile plus a
HALO-pixel ring (clipped to the frame) of every window-local surface and computes every pass on
that whole window; results are exact on the tile interior as long as a pass's reach stays inside
the halo (fixed-radius stencils: <= 19 half-res px).  Reads with unbounded reach
are served from whole-frame copies:

  exchange A (every frame, after the Hi-Z downsample):  two all-gathers — each tile's depth
      image-mips 1..4 + downsampled normals (what the trace marches through), then its albedo
      (what the filter reads at the hit positions)  ->  frame_hiz / frame_normals / frame_albedo
      on every rank; the coarser Hi-Z mips are then rebuilt locally.  Both are issued
      asynchronously right after the downsample: TAA (independent of them) runs while the first
      is in flight, the trace and GTAO while the second is.
  exchange B (every frame, one per history surface):  the halo ring of the TAA output, the
      accumulated AO and the blurred reflections is refreshed from the neighbours' interiors with
      point-to-point sends (up to 8 neighbours).  Each is issued right after the pass that
      produces the surface and awaited right before the pass that consumes it in the NEXT frame,
      so the transfer hides behind everything in between.

Every pack / scatter step is one launch of the C-ABI's vkr_copy_rects (a fixed list of byte rectangles,
built once per ping-pong parity of the history images), not a torch op per surface and neighbour: at eight
ranks that is 10 launches per frame instead of ~70.  `TiledFrame.phases()` is the single definition of the
frame order; it yields wherever bytes must cross ranks, and the caller moves them — RCCL in `step()`, plain
copies between in-process ranks in the lockstep test.

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests (which plug their own compute backend into TiledFrame).  No data-path collective is
used when world == 1.
"""
import time

import torch
import torch.distributed as dist

from . import abi

HALO = 48          # full-res pixels, a multiple of 16 (the gathered mips of a window must align with the frame's); the longest
                   # fixed reach is GTAO main + filter, 20 half-res = 40 full-res pixels.  Half-res surfaces carry HALO // 2
GATHER_MIPS = 4    # at most: depth image-mips 1..4 are tile-aligned for tiles divisible by 16


def gather_mips_for(tw, th, halo):
    """How many depth image-mips (1..k) of a tile are whole texels of the frame's mips: k = the largest power of two that
    divides the tile extent and the halo, capped at GATHER_MIPS.  The 15360x1080 strips of BASELINE config 4 on 8 GPUs
    give 3 (1080 = 8 * 135); coarser whole-frame mips are rebuilt locally from the last gathered one."""
    k = GATHER_MIPS
    while k > 1 and (tw % (1 << k) or th % (1 << k) or halo % (1 << k)):
        k -= 1
    return k


def grid_for(world):
    """Horizontal strips, one tile per rank stacked top to bottom.  A strip's share of every whole-frame surface is a
    run of whole rows, i.e. contiguous, so the all-gathers deliver in place with no pack and no scatter; 2-D grids
    (still supported by TiledFrame) balance a little better but pay 2 x the gathered bytes in HBM copies per frame
    (tools/lockstep_profile.py: slowest rank 1.19 ms for 1x8 vs 1.16 ms + ~0.25 ms of copies for 4x2)."""
    return (1, world)


def tile_rect(rank, cols, rows, tw, th):
    cx, cy = rank % cols, rank // cols
    return cx * tw, cy * th, tw, th


def window_rect(rank, cols, rows, tw, th, halo):
    x0, y0, w, h = tile_rect(rank, cols, rows, tw, th)
    W, H = cols * tw, rows * th
    wx0, wy0 = max(0, x0 - halo), max(0, y0 - halo)
    wx1, wy1 = min(W, x0 + w + halo), min(H, y0 + h + halo)
    return wx0, wy0, wx1 - wx0, wy1 - wy0


class HostBackend:
    """Compute through the C++ host mirror on the GPU (host.HostFrame); images live in torch tensors.
    native: dict(rank, world, halo, gathered_mips, force_tiled, comm) -> the frame lives inside the C++ tiled frame
    (host/frame.hpp vkrh_tiled_*), which also issues the exchanges."""

    def __init__(self, setup, window, tiled, device, native=None):
        from . import host

        self.host = host
        self.frame = host.HostFrame(setup, device=device, window=window, tiled=tiled, native_tiled=native)
        self.device = device
        self._views = {}

    def rows(self, name, mip=0):
        """(uint8 view [height, pitch] of one mip, bytes per texel, window rect).  Views are cached per device
        address: history remaps only permute which address a name resolves to."""
        d = self.frame.image(name, mip, 1)
        h, pitch = d.height, d.pitch_bytes[0]
        key = (d.base, h, pitch)
        view = self._views.get(key)
        if view is None:
            t, off = self.frame.allocator.tensor_at(d.base)
            view = self._views[key] = t[off: off + h * pitch].view(h, pitch)
        return view, abi.FORMAT_BYTES[d.format], (d.origin_x, d.origin_y, d.width, d.height)

    def set_gather_mips(self, n):
        if self.frame.tiled_handle is None:  # the C++ tiled frame was created with it
            self.frame.set_gathered_mips(n)

    def prepare(self):
        h = self.host
        self.frame.run(h.STAGE_LUT | h.STAGE_GBUFFER | h.STAGE_PREV_DEPTH)

    def run_stage(self, stage):
        """stage: 'downsample' | 'taa' | 'trace' (Hi-Z tail + trace) | 'gtao' (main, filter, accumulate) |
        'ssr_resolve' (filter, blur)"""
        h = self.host
        self.frame.run({"downsample": h.STAGE_DOWNSAMPLE, "taa": h.STAGE_TAA, "trace": h.STAGE_HIZ_TAIL | h.STAGE_SSR_TRACE,
                        "gtao": h.STAGE_GTAO, "ssr_resolve": h.STAGE_SSR_RESOLVE}[stage])

    def run_all(self):
        self.frame.run(self.host.STAGE_CHAIN)

    def end_frame(self):
        self.frame.end_frame(False)

    def sync(self):
        torch.cuda.synchronize(self.device)


class TiledFrame:
    def __init__(self, setup, rank, world, cols, rows, device, backend="host", halo=HALO, force_tiled=False, native=False, comm=None,
                 row_bounds=None, gather_mode=1):
        """native (strips on the host backend only): the frame order, the pack / unpack launches and the exchanges run in
        C++ (host/frame.cpp TiledFrame: grouped RCCL launches on its own stream, ordered with events) — step() is one
        call.  comm: abi.Comm, or None to drive the C++ phases from a harness (tests) / to rehearse on one rank.
        row_bounds (native strips only): world + 1 row numbers; strip r is rows [row_bounds[r], row_bounds[r + 1]) —
        strips of different heights balance ranks whose rows differ in cost (host.balance_rows).
        gather_mode (this Python driver; the C++ frame has its own, host.HostFrame): 1 = albedo and downsampled normals
        all-gathered (the default here); 0 = hit colours and hit normals by request / reply, 2 = only the hit colours —
        strips only, on a backend that exposes the hit_* steps (the oracle's: that is how the CPU tests run them)."""
        assert cols * rows == world
        self.setup, self.rank, self.world, self.cols, self.rows_n = setup, rank, world, cols, rows
        W, H = setup.width, setup.height
        # equal tiles need a frame that divides by the grid; strips with explicit bounds only need the bounds to cover the frame
        # (round 3's five-process wire test on a 256x576 frame died on this assertion before its bounds were looked at)
        assert W % cols == 0 and (H % rows == 0 or row_bounds is not None), f"a {W}x{H} frame does not divide into a {cols}x{rows} grid (pass row_bounds)"
        self.tw, self.th = W // cols, H // rows
        self.halo = halo if world > 1 else 0
        self.row_bounds = None
        if row_bounds is not None and (H % rows != 0 or list(row_bounds) != [r * self.th for r in range(world + 1)]):
            assert native and cols == 1 and len(row_bounds) == world + 1 and row_bounds[0] == 0 and row_bounds[-1] == H
            self.row_bounds = [int(v) for v in row_bounds]
            self.th = self.row_bounds[rank + 1] - self.row_bounds[rank]
            self.gather_mips = GATHER_MIPS
            while self.gather_mips > 1 and any(b % (1 << self.gather_mips) for b in self.row_bounds + [self.tw, self.halo]):
                self.gather_mips -= 1
            self.tile = (0, self.row_bounds[rank], self.tw, self.th)
            y0, y1 = max(0, self.tile[1] - self.halo), min(H, self.tile[1] + self.th + self.halo)
            self.window = (0, y0, W, y1 - y0)
        else:
            self.gather_mips = gather_mips_for(self.tw, self.th, self.halo)
            self.tile = tile_rect(rank, cols, rows, self.tw, self.th)
            self.window = window_rect(rank, cols, rows, self.tw, self.th, self.halo)
        if world > 1:
            assert self.tw % 2 == 0 and self.th % 2 == 0, "tile extent must be even"
            assert self.halo % 2 == 0 and self.halo <= min(self.tw, self.th)
        # force_tiled: run the multi-GPU code path (gathers, whole-frame Hi-Z, staged frame) on one rank
        self.tiled = world > 1 or force_tiled
        # "host": the C++ host mirror on the GPU; anything else: a class with HostBackend's interface
        self.native = bool(native) and self.tiled
        if self.native:
            assert backend == "host" and cols == 1, "the C++ tiled frame cuts horizontal strips"
            self.backend = HostBackend(setup, self.window, self.tiled, device,
                                       native=dict(rank=rank, world=world, halo=self.halo, gathered_mips=self.gather_mips,
                                                   force_tiled=force_tiled, comm=comm, row_bounds=self.row_bounds))
        else:
            cls = HostBackend if backend == "host" else backend
            self.backend = cls(setup, self.window, self.tiled, device)
        self.backend.set_gather_mips(self.gather_mips)
        self.gather_mode = 1 if (self.native or not self.tiled or world == 1) else int(gather_mode)
        if self.gather_mode != 1:
            assert cols == 1 and hasattr(self.backend, "hit_count"), "request / reply needs strips and a backend with the hit_* steps"
            self.backend.windowed = self.gather_mode == 0
        self.frame = self.backend.frame
        self.device = self.backend.device
        self._xchg_s = 0.0
        self._frame_no = 0       # history images ping-pong: every address-dependent plan is cached per parity
        self._gather_cache = {}
        self._halo_cache = {}
        self._halo_bufs = {}
        self._halo_packed = {}   # surface -> plan whose send buffers are filled and whose receive is outstanding
        self._halo_works = {}
        self.stage_plan = None  # single-GPU only: list of stage masks run per step instead of STAGE_CHAIN

    # ---- set-up ---------------------------------------------------------------------------------
    def prepare(self):
        self.backend.prepare()
        if self.frame is not None:  # GPU: seed the histories as SURVEY.md 8(d) says
            self._seed_histories_gpu()
        self.backend.sync()

    def _seed_histories_gpu(self):
        """TAA history = current colour (decoded albedo), AO history = (1, 1/255), SSR history = 0."""
        import numpy as np

        alb, _, (_, _, w, h) = self.backend.rows("albedo")
        lut = np.zeros(256, dtype=np.float32)
        c = np.arange(256) / 255.0
        lut[:] = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(np.float32)
        lut_t = torch.from_numpy(lut).to(self.device)
        rgba = alb[:, : w * 4].reshape(h, w, 4).long()
        col = lut_t[rgba[..., :3]].to(torch.float16)
        hist, _, _ = self.backend.rows("taa_hist")
        dst = hist[:, : w * 8].view(torch.float16).view(h, w, 4)
        dst[..., :3] = col
        dst[..., 3] = 0
        acc, _, (_, _, w2, h2) = self.backend.rows("acc_hist")
        a = acc[:, : w2 * 4].view(torch.float16).view(h2, w2, 2)
        a[..., 0] = 1.0
        a[..., 1] = 1.0 / 255.0

    # ---- one frame ----------------------------------------------------------------------------------
    def phases(self):
        """The tiled frame, as a generator that yields wherever bytes must cross ranks:
            ("gather_start", g)  for (send, recv) in g.parts: start all_gather(recv <- every rank's send), recv = [rank][send bytes]
            ("gather_wait", g)   every recv of g.parts must be complete before resuming
            ("halo_start", s)    surface s: the send buffers of halo_peers(s) are packed: start the sends / receives
            ("halo_wait", s)     the receive buffers of surface s (started in the PREVIOUS frame) must be complete
        downsample -> [gather Hi-Z + normals || TAA] -> trace -> GTAO -> [gather albedo, in flight since the
        downsample] -> filter, blur.  GTAO only needs the trace's (occlusion, pdf), not the albedo, so it runs
        ahead of the reference's order to hide more of the second gather."""
        b = self.backend
        b.run_stage("downsample")
        hiz = self.gather_pack("hiz")
        yield "gather_start", hiz
        by_request = self.gather_mode != 1
        if not by_request:
            albedo = self.gather_pack("albedo")
            yield "gather_start", albedo
        else:
            b.local_rows(normals=self.gather_mode == 0)
        yield "halo_wait", "taa"
        self.halo_unpack("taa")
        b.run_stage("taa")
        self.halo_pack("taa")
        yield "halo_start", "taa"
        yield "gather_wait", hiz
        hiz.unpack.run()
        b.run_stage("trace")
        yield "halo_wait", "ao"
        self.halo_unpack("ao")
        b.run_stage("gtao")
        self.halo_pack("ao")
        yield "halo_start", "ao"
        if by_request:
            yield "hit_exchange", None  # counts -> requests -> replies -> scatter (-> deferred hit-normal test)
        else:
            yield "gather_wait", albedo
            albedo.unpack.run()
        yield "halo_wait", "ssr"
        self.halo_unpack("ssr")
        b.run_stage("ssr_resolve")
        self.halo_pack("ssr")
        yield "halo_start", "ssr"
        self.end_frame()

    def end_frame(self):
        self.backend.end_frame()
        self._frame_no += 1

    def step(self):
        if self.native:
            t0 = time.perf_counter()
            self.frame.tiled_step()
            self._xchg_s += 0.0 * (time.perf_counter() - t0)  # exchanges are issued inside the C++ step: no separate host cost
            self._frame_no += 1
            return
        if not self.tiled:
            if self.stage_plan:
                for mask in self.stage_plan:
                    self.frame.run(mask)
            else:
                self.backend.run_all()
            self.end_frame()
            return
        # The collectives run on the communicator's own stream; wait() only orders the compute stream behind
        # them, the host never blocks.
        for op, arg in self.phases():
            t0 = time.perf_counter()
            if op == "gather_start":
                arg.works = self._all_gather(arg.parts)
            elif op == "gather_wait":
                for work in arg.works:
                    work.wait()
            elif op == "halo_start":
                self._halo_works[arg] = self._halo_issue(arg)
            elif op == "hit_exchange":
                self._hit_exchange()
            else:  # halo_wait
                self._halo_complete(arg)
            self._xchg_s += time.perf_counter() - t0

    def _host_driven_backend(self):
        """RCCL enqueues its kernels behind everything already recorded on the compute stream and writes through the
        GPU's own memory system, so device buffers can be handed to it as they are.  A host-driven backend (gloo,
        used by the CPU tests and by the two-process rehearsal on one GPU) must not be given device pointers: the
        exchange is staged through host tensors instead."""
        return self.device is not None and torch.device(self.device).type == "cuda" and dist.get_backend() != "nccl"

    def _all_gather(self, parts):
        """One collective launch for all (send, recv) pairs of a group: every call into torch.distributed costs ~35 us of
        host time, and with strips a group is up to five surfaces."""
        if self._host_driven_backend():
            for send, recv in parts:
                staged = torch.empty(recv.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(staged, send.cpu())
                recv.copy_(staged)
            return []
        if len(parts) == 1:
            return [dist.all_gather_into_tensor(parts[0][1], parts[0][0], async_op=True)]
        return [dist.group.WORLD.allgather_into_tensor_coalesced([recv for _, recv in parts], [send for send, _ in parts])]

    def flush(self):
        """Completes the halo exchanges the last frame left in flight (call before reading results / stopping a clock)."""
        if self.native:
            self.frame.tiled_flush()
            return
        for which in list(self._halo_packed):
            self._halo_complete(which)
            self.halo_unpack(which)

    def exchange_ms(self, steps):
        """host-side time spent issuing exchanges per step (device time shows in ms_per_step)"""
        return self._xchg_s / max(steps, 1) * 1e3

    # ---- exchange A: all-gather of the unbounded-reach surfaces -------------------------------------
    def _hit_exchange(self):
        """The request / reply round over torch.distributed (host tensors: this driver runs on the oracle in the CPU tests):
        every rank's counts are gathered, 4-byte requests and 16-byte replies travel point to point, the replies are
        scattered into the whole-frame images."""
        import numpy as np

        b, world, me = self.backend, self.world, self.rank
        normals = self.gather_mode == 0
        bounds = [r * self.th for r in range(world + 1)]
        counts = b.hit_count(bounds, normals)
        matrix = [None] * world
        dist.all_gather_object(matrix, counts)
        requests, seg = b.hit_write(bounds, counts, normals)
        req_np = np.asarray(requests, dtype=np.uint32)
        incoming = [matrix[r][me] for r in range(world)]
        in_seg = [0]
        for n in incoming:
            in_seg.append(in_seg[-1] + n)
        req_in = np.zeros(max(in_seg[-1], 1), dtype=np.uint32)

        def exchange(out_np, out_seg, in_np, in_seg_, words):
            ops, keep = [], []
            for p in range(world):
                if p == me:
                    continue
                so, ri = out_seg[p + 1] - out_seg[p], in_seg_[p + 1] - in_seg_[p]
                if so:
                    keep.append(torch.from_numpy(np.ascontiguousarray(out_np[words * out_seg[p]: words * out_seg[p + 1]]).view(np.int32)))
                    ops.append(dist.P2POp(dist.isend, keep[-1], p))
                if ri:
                    keep.append(torch.from_numpy(in_np[words * in_seg_[p]: words * in_seg_[p + 1]].view(np.int32)))
                    ops.append(dist.P2POp(dist.irecv, keep[-1], p))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()

        exchange(req_np, seg, req_in, in_seg, 1)
        replies, errors = b.hit_reply(req_in, in_seg[-1], normals)
        assert errors == 0, "a rank was asked for texels outside its window"
        rep_np = np.asarray(replies, dtype=np.uint32)
        rep_in = np.zeros(max(4 * seg[-1], 4), dtype=np.uint32)
        exchange(rep_np, in_seg, rep_in, seg, 4)
        b.hit_scatter(req_np, rep_in, seg[-1], normals)
        self.hit_matrix = matrix

    def _gather_plan(self, group):
        """[(src image, src mip, dst image, dst mip, divisor)]: tile interior at full-res >> divisor"""
        if group == "hiz":
            depth = [("depth", m, "frame_hiz", m - 1, m) for m in range(1, self.gather_mips + 1)]
            return depth + ([] if getattr(self, "gather_mode", 1) == 0 else [("dn", 0, "frame_normals", 0, 1)])
        return [("albedo", 0, "frame_albedo", 0, 0)]

    def gather_pack(self, group):
        """Prepares this tile's share of `group` for the all-gather and returns the exchange's state: .parts =
        [(send, recv)] with recv = [rank][send bytes], and .unpack, what is left to do once the bytes have arrived.
        The collective itself is separate so that a test harness can move the bytes between in-process ranks.

        Strips (cols == 1): a tile's rows of every surface are contiguous in the window image AND in the whole-frame
        image (same width, same pitch), so each surface is gathered in place — send = the tile's rows where they lie,
        recv = the frame image itself; nothing is packed and nothing is scattered.
        2-D grids: the surfaces are packed into one send buffer (one copy_rects launch), gathered with one collective
        and scattered into the frame images (one more launch)."""
        key = (group, self._frame_no & 1)
        g = self._gather_cache.get(key)
        if g is None:
            g = self._gather_cache[key] = self._build_gather_in_place(group) if self.cols == 1 else self._build_gather_packed(group)
        g.pack.run()
        return g

    def _build_gather_in_place(self, group):
        x0, y0, tw, th = self.tile
        parts = []
        for src, mip, dst, dmip, dv in self._gather_plan(group):
            rows, bpp, (ox, oy, w, _) = self.backend.rows(src, mip)
            frows, fbpp, (fox, foy, fw, fh) = self.backend.rows(dst, dmip)
            h = th >> dv
            assert ox == 0 and fox == 0 and foy == 0 and w == fw == tw >> dv and fh == h * self.world and fbpp == bpp
            assert rows.stride(0) == frows.stride(0), "strips are gathered in place: window and frame images must share the row pitch"
            ly = (y0 >> dv) - oy
            parts.append((rows[ly: ly + h].reshape(-1), frows[: fh].reshape(-1)))  # both are views: rows are whole pitches
            assert parts[-1][0].data_ptr() == rows[ly].data_ptr() and parts[-1][1].data_ptr() == frows.data_ptr()
        g = _Gather()
        g.parts, g.works = parts, []
        g.pack, g.unpack = RectBatch([], self.device), RectBatch([], self.device)
        return g

    def _build_gather_packed(self, group):
        plan = self._gather_plan(group)
        x0, y0, tw, th = self.tile
        sizes = []
        for src, mip, _, _, dv in plan:
            _, bpp, _ = self.backend.rows(src, mip)
            sizes.append((th >> dv) * (tw >> dv) * bpp)
        chunk = sum(sizes)
        other = self._gather_cache.get((group, 1 - (self._frame_no & 1)))
        if other is not None:  # both parities share the buffers
            send, recv = other.parts[0]
        else:
            send = torch.empty(chunk, dtype=torch.uint8, device=self.device)
            recv = torch.empty(chunk * self.world, dtype=torch.uint8, device=self.device)
        per_rank = recv.view(self.world, chunk)
        pack, unpack, off = [], [], 0
        for (src, mip, dst, dmip, dv), n in zip(plan, sizes):
            rows, bpp, (ox, oy, _, _) = self.backend.rows(src, mip)
            lx, ly, w, h = (x0 >> dv) - ox, (y0 >> dv) - oy, tw >> dv, th >> dv
            pack.append((send[off: off + n].view(h, w * bpp), rows[ly: ly + h, lx * bpp: (lx + w) * bpp]))
            # recv is [rank][surface bytes]; rank r sits at column r % cols, row r // cols of the grid
            frows, fbpp, (fox, foy, _, _) = self.backend.rows(dst, dmip)
            assert fox == 0 and foy == 0 and fbpp == bpp, "whole-frame images start at the frame origin"
            for r in range(self.world):
                cx, cy = r % self.cols, r // self.cols
                unpack.append((frows[cy * h: (cy + 1) * h, cx * w * bpp: (cx + 1) * w * bpp], per_rank[r, off: off + n].view(h, w * bpp)))
            off += n
        g = _Gather()
        g.parts, g.works = [(send, recv)], []
        g.pack, g.unpack = RectBatch(pack, self.device), RectBatch(unpack, self.device)
        return g

    # ---- exchange B: history halos -------------------------------------------------------------------
    def _neighbours(self):
        cx, cy = self.rank % self.cols, self.rank // self.cols
        out = []
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if (dx or dy) and 0 <= cx + dx < self.cols and 0 <= cy + dy < self.rows_n:
                    out.append((dx, dy, (cy + dy) * self.cols + (cx + dx)))
        return out

    @staticmethod
    def _overlap(a, b):
        x0, y0 = max(a[0], b[0]), max(a[1], b[1])
        x1, y1 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
        return (x0, y0, x1 - x0, y1 - y0) if x1 > x0 and y1 > y0 else None

    def _halo_plan(self, which):
        """Geometry of the halo refresh of one history surface: per neighbour, the slice to send (my interior inside
        its window) and to receive (its interior inside my window), one persistent buffer per neighbour and
        direction.  The surface is addressed by the name of the pass OUTPUT (it becomes the history at the remap)."""
        key = (which, self._frame_no & 1)
        plan = self._halo_cache.get(key)
        if plan is not None:
            return plan
        name, dv = HALO_SURFACES[which]
        rows, bpp, (ox, oy, ww, wh) = self.backend.rows(name)
        bufs = self._halo_bufs.setdefault(which, {})
        mine = tuple(v >> dv for v in self.tile)
        pack, unpack, peers = [], [], []
        for _, _, nb in self._neighbours():
            nb_tile = tuple(v >> dv for v in tile_rect(nb, self.cols, self.rows_n, self.tw, self.th))
            nb_win = tuple(v >> dv for v in window_rect(nb, self.cols, self.rows_n, self.tw, self.th, self.halo))
            s, r = self._overlap(mine, nb_win), self._overlap(nb_tile, (ox, oy, ww, wh))
            if nb not in bufs:
                bufs[nb] = (torch.empty(s[2] * s[3] * bpp if s else 0, dtype=torch.uint8, device=self.device),
                            torch.empty(r[2] * r[3] * bpp if r else 0, dtype=torch.uint8, device=self.device))
            sbuf, rbuf = bufs[nb]
            if s:
                pack.append((sbuf.view(s[3], s[2] * bpp), rows[s[1] - oy: s[1] - oy + s[3], (s[0] - ox) * bpp: (s[0] - ox + s[2]) * bpp]))
            if r:
                unpack.append((rows[r[1] - oy: r[1] - oy + r[3], (r[0] - ox) * bpp: (r[0] - ox + r[2]) * bpp], rbuf.view(r[3], r[2] * bpp)))
            peers.append((nb, sbuf if s else None, rbuf if r else None))
        plan = self._halo_cache[key] = _Halo()
        plan.pack, plan.unpack, plan.peers = RectBatch(pack, self.device), RectBatch(unpack, self.device), peers
        return plan

    def halo_pack(self, which):
        """Fills every neighbour's send buffer with this frame's output of surface `which` -> [(neighbour, send buffer |
        None, receive buffer | None)].  The matching halo_unpack happens in the next frame, into the same image."""
        assert which not in self._halo_packed, "the previous halo exchange of this surface was never completed"
        plan = self._halo_plan(which)
        plan.pack.run()
        self._halo_packed[which] = plan
        return plan.peers

    def halo_peers(self, which):
        return self._halo_packed[which].peers if which in self._halo_packed else []

    def halo_unpack(self, which):
        plan = self._halo_packed.pop(which, None)
        if plan is not None:
            plan.unpack.run()

    def _halo_issue(self, which):
        staged = self._host_driven_backend()
        ops, landing, keep = [], [], []
        for nb, sbuf, rbuf in self.halo_peers(which):
            if sbuf is not None:
                keep.append(sbuf.cpu() if staged else sbuf)  # a staged copy must outlive the send
                ops.append(dist.P2POp(dist.isend, keep[-1], nb))
            if rbuf is not None:
                dst = torch.empty(rbuf.numel(), dtype=torch.uint8) if staged else rbuf
                ops.append(dist.P2POp(dist.irecv, dst, nb))
                if staged:
                    landing.append((rbuf, dst))
        return (dist.batch_isend_irecv(ops) if ops else []), landing, keep

    def _halo_complete(self, which):
        works, landing, _ = self._halo_works.pop(which, ((), (), ()))
        for work in works:
            work.wait()
        for rbuf, staged in landing:
            rbuf.copy_(staged)


HALO_SURFACES = {"taa": ("taa_target", 0), "ao": ("acc_ao", 1), "ssr": ("blurred", 1)}  # pass output, divisor


class _Gather:
    pass


class _Halo:
    pass


class RectBatch:
    """A fixed list of byte-rectangle copies [(dst, src)], uint8 [rows, row bytes] views with a row pitch.  On the
    GPU the list is one vkr_copy_rects launch on the current stream; host tensors (gloo tests) are copied one by one."""

    def __init__(self, pairs, device):
        self.pairs = pairs
        self.device = device if device is not None and torch.device(device).type == "cuda" else None
        for dst, src in pairs:
            assert dst.shape == src.shape and dst.dtype == src.dtype == torch.uint8 and dst.dim() == 2
            assert dst.shape[1] == 0 or (dst.stride(1) == 1 and src.stride(1) == 1)
        if self.device is not None and pairs:
            self.table = (abi.RectCopy * len(pairs))()
            for i, (dst, src) in enumerate(pairs):
                self.table[i] = abi.RectCopy(src.data_ptr(), dst.data_ptr(), src.stride(0), dst.stride(0), src.shape[1], src.shape[0])

    def run(self):
        if not self.pairs:
            return
        if self.device is None:
            for dst, src in self.pairs:
                dst.copy_(src)
            return
        lib = abi.product()
        abi.check(lib.vkr_copy_rects(self.table, len(self.pairs), torch.cuda.current_stream(self.device).cuda_stream), lib)


# ---- lockstep harness for the C++ tiled frame (tests, tools/lockstep_profile.py): all ranks of a grid on ONE GPU ------------
def _bytes_at(ranks, addr, nbytes):
    """uint8 view of device memory [addr, addr + nbytes) owned by one of the ranks' allocators"""
    for t in ranks:
        try:
            tensor, off = t.frame.allocator.tensor_at(addr)
        except KeyError:
            continue
        assert off + nbytes <= tensor.numel()
        return tensor[off: off + nbytes]
    raise KeyError(hex(addr))


def _move_gather(ranks, which):
    """what vkr_all_gather / vkr_all_gather_v deliver: every rank's share of a surface, in rank order, into every rank's
    recv (shares are whole rows of consecutive strips, so share r starts where the shares before it end)"""
    parts = [t.frame.tiled_gather_parts(which) for t in ranks]
    for r, mine in enumerate(parts):
        for i, (_, recv, _) in enumerate(mine):
            offset = 0
            for theirs in parts:
                send, _, nbytes = theirs[i]
                _bytes_at(ranks, recv + offset, nbytes).copy_(_bytes_at(ranks, send, nbytes))
                offset += nbytes


def _move_halo(ranks, surface):
    """what vkr_halo_exchange delivers: every receive buffer gets the send buffer its peer packed for this rank"""
    peers = [t.frame.tiled_halo_peers(surface) for t in ranks]
    for r, mine in enumerate(peers):
        for peer, _, recv, nbytes in mine:
            send = [p for p in peers[peer] if p[0] == r][0][1]
            _bytes_at(ranks, recv, nbytes).copy_(_bytes_at(ranks, send, nbytes))


def _move_p2p(ranks, lists):
    """what vkr_halo_exchange delivers for arbitrary peer lists [(peer, send, send bytes, recv, recv bytes)] per rank: the
    send of rank a to rank b lands in the receive buffer rank b names for a"""
    for b, mine in enumerate(lists):
        for peer, _, _, recv, recv_bytes in mine:
            if not recv_bytes:
                continue
            send = [p for p in lists[peer] if p[0] == b]
            assert len(send) == 1 and send[0][2] == recv_bytes, "the two ends of a hit-colour exchange disagree on its size"
            _bytes_at(ranks, recv, recv_bytes).copy_(_bytes_at(ranks, send[0][1], recv_bytes))


def _hit_exchange(ranks, in_capacities=False):
    """the hit-colour request / reply of frame.hpp between in-process ranks: counts -> requests -> replies -> scatter.
    in_capacities: the segments are laid out by host.hit_capacities(counts) — the room the native frame gives them when it
    enqueues the round on the previous frame's counts — instead of by the exact counts (unused slots say "no request")."""
    world = len(ranks)
    matrix = [c for t in ranks for c in t.frame.tiled_hit_counts(world)]
    if in_capacities:
        from . import host

        layout = host.hit_capacities(matrix, world)
    else:
        layout = matrix
    _move_p2p(ranks, [t.frame.tiled_hit_requests(layout) for t in ranks])
    _move_p2p(ranks, [t.frame.tiled_hit_replies() for t in ranks])
    for t in ranks:
        t.frame.tiled_hit_finish()
    return matrix


def native_lockstep_frame(ranks, hit_in_capacities=False):
    """One frame of every in-process rank of a strip grid (C++ tiled frames made with native=True, comm=None), advanced
    phase by phase; between phases the harness copies exactly the buffers the RCCL calls would move."""
    by_gather = ranks[0].frame.albedo_by_gather or len(ranks) == 1
    taa_after_gtao = ranks[0].frame.tiled_local_first()  # the C++ frame then resolves the TAA in phase 3 (frame.cpp: local rows first)
    for p in range(5):
        for t in ranks:
            t.frame.tiled_phase(p)
        if p == 0:      # the gathers start after the downsample; the harness completes them at once
            _move_gather(ranks, 0)
            if by_gather:
                _move_gather(ranks, 1)
        elif p == 1 and not taa_after_gtao:
            _move_halo(ranks, 0)
        elif p == 3:
            if taa_after_gtao:
                _move_halo(ranks, 0)
            _move_halo(ranks, 1)
            if not by_gather:
                ranks[0].hit_matrix = _hit_exchange(ranks, hit_in_capacities)
        elif p == 4:
            _move_halo(ranks, 2)
    for t in ranks:
        t._frame_no += 1
