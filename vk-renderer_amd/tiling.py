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
  exchange B (every frame, after the history remaps):  the halo ring of the three history
      surfaces (TAA, accumulated AO, blurred reflections) is refreshed from the neighbours'
      interiors with point-to-point sends (up to 8 neighbours).

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests (which plug their own compute backend into TiledFrame).  No data-path collective is
used when world == 1.
"""
import time

import torch
import torch.distributed as dist

from . import abi

HALO = 64          # full-res pixels; half-res surfaces carry HALO // 2
GATHER_MIPS = 4    # depth image-mips 1..4 are tile-aligned for tiles divisible by 16


def grid_for(world):
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(world, (world, 1))


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
    """Compute through the C++ host mirror on the GPU (host.HostFrame); images live in torch tensors."""

    def __init__(self, setup, window, tiled, device):
        from . import host

        self.host = host
        self.frame = host.HostFrame(setup, device=device, window=window, tiled=tiled)
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
    def __init__(self, setup, rank, world, cols, rows, device, backend="host", halo=HALO, force_tiled=False):
        assert cols * rows == world
        self.setup, self.rank, self.world, self.cols, self.rows_n = setup, rank, world, cols, rows
        W, H = setup.width, setup.height
        assert W % cols == 0 and H % rows == 0
        self.tw, self.th = W // cols, H // rows
        self.halo = halo if world > 1 else 0
        if world > 1:
            assert self.tw % (1 << GATHER_MIPS) == 0 and self.th % (1 << GATHER_MIPS) == 0, "tile must be divisible by 16"
            assert self.halo % 2 == 0 and self.halo <= min(self.tw, self.th)
        self.tile = tile_rect(rank, cols, rows, self.tw, self.th)
        self.window = window_rect(rank, cols, rows, self.tw, self.th, self.halo)
        # force_tiled: run the multi-GPU code path (gathers, whole-frame Hi-Z, staged frame) on one rank
        self.tiled = world > 1 or force_tiled
        # "host": the C++ host mirror on the GPU; anything else: a class with HostBackend's interface
        cls = HostBackend if backend == "host" else backend
        self.backend = cls(setup, self.window, self.tiled, device)
        self.frame = self.backend.frame
        self.device = self.backend.device
        self._xchg_s = 0.0
        self._gather_buf = {}
        self._halo_cache = None
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
    def step(self):
        if not self.tiled:
            if self.stage_plan:
                for mask in self.stage_plan:
                    self.frame.run(mask)
            else:
                self.backend.run_all()
            self.backend.end_frame()
            return
        # downsample -> [gather Hi-Z + normals || TAA] -> trace -> GTAO -> [gather albedo, in flight since
        # the downsample] -> filter, blur.  GTAO only needs the trace's (occlusion, pdf), not the albedo,
        # so it runs ahead of the reference's order to hide more of the second gather.  The collectives
        # run on the communicator's own stream; wait() only orders the compute stream behind them.
        self.backend.run_stage("downsample")
        t0 = time.perf_counter()
        pending_hiz = self.gather_start("hiz")
        pending_albedo = self.gather_start("albedo")
        self._xchg_s += time.perf_counter() - t0
        self.backend.run_stage("taa")
        t0 = time.perf_counter()
        self.gather_finish(pending_hiz)
        self._xchg_s += time.perf_counter() - t0
        self.backend.run_stage("trace")
        self.backend.run_stage("gtao")
        t0 = time.perf_counter()
        self.gather_finish(pending_albedo)
        self._xchg_s += time.perf_counter() - t0
        self.backend.run_stage("ssr_resolve")
        self.backend.end_frame()
        t0 = time.perf_counter()
        self.exchange_history_halos()
        self._xchg_s += time.perf_counter() - t0

    def exchange_ms(self, steps):
        """host-side time spent issuing exchanges per step (device time shows in ms_per_step)"""
        return self._xchg_s / max(steps, 1) * 1e3

    # ---- exchange A: all-gather of the unbounded-reach surfaces -------------------------------------
    def _gather_plan(self, group):
        """[(src image, src mip, dst image, dst mip, divisor)]: tile interior at full-res >> divisor"""
        if group == "hiz":
            return [("depth", m, "frame_hiz", m - 1, m) for m in range(1, GATHER_MIPS + 1)] + [("dn", 0, "frame_normals", 0, 1)]
        return [("albedo", 0, "frame_albedo", 0, 0)]

    def gather_pack(self, group):
        """Packs this tile's share of `group` into the send buffer -> (send, recv, plan, sizes, chunk).  The collective
        itself is separate so that a test harness can move the bytes between in-process ranks instead."""
        plan = self._gather_plan(group)
        x0, y0, tw, th = self.tile
        sizes = []
        for src, mip, _, _, dv in plan:
            _, bpp, _ = self.backend.rows(src, mip)
            sizes.append((th >> dv) * (tw >> dv) * bpp)
        chunk = sum(sizes)
        if group not in self._gather_buf:
            self._gather_buf[group] = (torch.empty(chunk, dtype=torch.uint8, device=self.device),
                                       torch.empty(chunk * self.world, dtype=torch.uint8, device=self.device))
        send, recv = self._gather_buf[group]
        off = 0
        for (src, mip, _, _, dv), n in zip(plan, sizes):
            rows, bpp, (ox, oy, _, _) = self.backend.rows(src, mip)
            lx, ly, w, h = (x0 >> dv) - ox, (y0 >> dv) - oy, tw >> dv, th >> dv
            send[off: off + n].view(h, w * bpp).copy_(rows[ly: ly + h, lx * bpp: (lx + w) * bpp])
            off += n
        return send, recv, plan, sizes, chunk

    def gather_start(self, group):
        """Packs and issues the all-gather asynchronously."""
        send, recv, plan, sizes, chunk = self.gather_pack(group)
        work = dist.all_gather_into_tensor(recv, send, async_op=True)
        return work, plan, sizes, chunk, recv

    def gather_finish(self, pending):
        """Orders the compute stream behind the collective and scatters the tiles into the frame images."""
        work, plan, sizes, chunk, recv = pending
        work.wait()
        self.gather_unpack(plan, sizes, chunk, recv)

    def gather_unpack(self, plan, sizes, chunk, recv):
        # one strided copy per surface: recv is [rank = (row, col)][surface bytes]; the frame image is
        # [row][y][col][x bytes] (rank r sits at column r % cols, row r // cols of the grid)
        per_rank = recv.view(self.world, chunk)
        off = 0
        for (_, _, dst, dmip, dv), n in zip(plan, sizes):
            rows, bpp, (ox, oy, _, _) = self.backend.rows(dst, dmip)
            assert ox == 0 and oy == 0, "whole-frame images start at the frame origin"
            w, h = self.tw >> dv, self.th >> dv
            src = per_rank[:, off: off + n].view(self.rows_n, self.cols, h, w * bpp)
            out = rows[: self.rows_n * h, : self.cols * w * bpp].unflatten(0, (self.rows_n, h)).unflatten(2, (self.cols, w * bpp))
            out.copy_(src.permute(0, 2, 1, 3))
            off += n

    def exchange_gather(self):
        """Both groups back to back (kept for callers that do not overlap)."""
        for group in ("hiz", "albedo"):
            self.gather_finish(self.gather_start(group))

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

    def _halo_plan(self):
        """Static geometry of exchange B: per neighbour, the slices of each history surface to send (my
        interior inside its window) and to receive (its interior inside my window), packed back to back
        in one persistent buffer per direction."""
        if self._halo_cache is not None:
            return self._halo_cache
        plan = []
        for dx, dy, nb in self._neighbours():
            send, recv, sbytes, rbytes = [], [], 0, 0
            for name, dv in (("taa_hist", 0), ("acc_hist", 1), ("blurred_hist", 1)):
                _, bpp, (ox, oy, ww, wh) = self.backend.rows(name)
                mine = tuple(v >> dv for v in self.tile)
                nb_tile = tuple(v >> dv for v in tile_rect(nb, self.cols, self.rows_n, self.tw, self.th))
                nb_win = tuple(v >> dv for v in window_rect(nb, self.cols, self.rows_n, self.tw, self.th, self.halo))
                s = self._overlap(mine, nb_win)
                if s:
                    send.append((name, s[1] - oy, s[3], (s[0] - ox) * bpp, s[2] * bpp, sbytes))
                    sbytes += s[3] * s[2] * bpp
                r = self._overlap(nb_tile, (ox, oy, ww, wh))
                if r:
                    recv.append((name, r[1] - oy, r[3], (r[0] - ox) * bpp, r[2] * bpp, rbytes))
                    rbytes += r[3] * r[2] * bpp
            plan.append((nb, send, torch.empty(max(sbytes, 1), dtype=torch.uint8, device=self.device),
                         recv, torch.empty(max(rbytes, 1), dtype=torch.uint8, device=self.device)))
        self._halo_cache = plan
        return plan

    def halo_pack(self):
        """Fills every neighbour's send buffer -> the plan [(neighbour, send slices, send buffer, recv slices, recv buffer)]."""
        plan = self._halo_plan()
        rows = {name: self.backend.rows(name)[0] for name in ("taa_hist", "acc_hist", "blurred_hist")}
        for nb, send, sbuf, recv, rbuf in plan:
            for name, y, h, xb, wb, off in send:
                sbuf[off: off + h * wb].view(h, wb).copy_(rows[name][y: y + h, xb: xb + wb])
        return plan

    def halo_unpack(self, plan):
        rows = {name: self.backend.rows(name)[0] for name in ("taa_hist", "acc_hist", "blurred_hist")}
        for nb, send, sbuf, recv, rbuf in plan:
            for name, y, h, xb, wb, off in recv:
                rows[name][y: y + h, xb: xb + wb].copy_(rbuf[off: off + h * wb].view(h, wb))

    def exchange_history_halos(self):
        plan = self.halo_pack()
        ops = []
        for nb, send, sbuf, recv, rbuf in plan:
            if send:
                ops.append(dist.P2POp(dist.isend, sbuf, nb))
            if recv:
                ops.append(dist.P2POp(dist.irecv, rbuf, nb))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        self.halo_unpack(plan)
