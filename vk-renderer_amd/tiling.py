"""Frame tiling across the GPUs of one node.

The frame is cut into cols x rows tiles, one per rank.  Every rank holds its tile plus a
HALO-pixel ring (clipped to the frame) of every window-local surface and computes every pass on
that whole window; results are exact on the tile interior as long as a pass's reach stays inside
the halo (fixed-radius stencils: <= 19 half-res px).  Reads with unbounded reach
are served from whole-frame copies:

  exchange A (every frame, after the Hi-Z downsample):  all-gather of each tile's
      depth image-mips 1..4, downsampled normals and albedo  ->  frame_hiz / frame_normals /
      frame_albedo on every rank; the coarser Hi-Z mips are then rebuilt locally.
  exchange B (every frame, after the history remaps):  the halo ring of the three history
      surfaces (TAA, accumulated AO, blurred reflections) is refreshed from the neighbours'
      interiors with point-to-point sends (up to 8 neighbours).

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests (where the compute backend is the oracle).  No data-path collective is
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

    def rows(self, name, mip=0):
        d = self.frame.image(name, mip, 1)
        t, off = self.frame.allocator.tensor_at(d.base)
        h, pitch = d.height, d.pitch_bytes[0]
        return t[off: off + h * pitch].view(h, pitch), abi.FORMAT_BYTES[d.format], (d.origin_x, d.origin_y, d.width, d.height)

    def prepare(self):
        h = self.host
        self.frame.run(h.STAGE_LUT | h.STAGE_GBUFFER | h.STAGE_PREV_DEPTH)

    def run_pre(self):
        self.frame.run(self.host.STAGE_DOWNSAMPLE)

    def run_main(self):
        h = self.host
        self.frame.run(h.STAGE_HIZ_TAIL | h.STAGE_SSR | h.STAGE_GTAO | h.STAGE_TAA)

    def run_all(self):
        self.frame.run(self.host.STAGE_CHAIN)

    def end_frame(self):
        self.frame.end_frame(False)

    def sync(self):
        torch.cuda.synchronize(self.device)


class OracleBackend:
    """TESTS ONLY: the same driver over the CPU oracle (chain.PostFxChain, host memory)."""

    def __init__(self, setup, window, tiled, device=None):
        from .chain import PostFxChain

        self.chain = PostFxChain(setup.width, setup.height, backend="oracle", setup=setup, window=window, force_tiled=tiled)
        self.frame = None
        self.device = torch.device("cpu")

    def rows(self, name, mip=0):
        img = getattr(self.chain, name)
        from .images import mip_extent

        h, w = mip_extent(img.height, mip), mip_extent(img.width, mip)
        t = torch.from_numpy(img.host)[img.offset[mip]: img.offset[mip] + h * img.pitch[mip]].view(h, img.pitch[mip])
        return t, img.bpp, (img.origin[0] >> mip, img.origin[1] >> mip, w, h)

    def prepare(self):
        c = self.chain
        c.synth()
        c.build_prev_hiz()
        c.init_histories()
        c.preintegrate_pdf()

    def run_pre(self):
        self.chain.downsample()

    def run_main(self):
        c = self.chain
        if c.tiled:
            c.hiz_tail(GATHER_MIPS)
        c.ssr_trace(frame_random=c.frame_index % 16)
        c.ssr_filter()
        c.ssr_blur()
        c.gtao_main()
        c.gtao_filter()
        c.gtao_accumulate()
        c.taa()
        c.frame_index += 1

    def run_all(self):
        self.chain.frame()

    def end_frame(self):
        self.chain.swap_histories()

    def sync(self):
        pass


class TiledFrame:
    def __init__(self, setup, rank, world, cols, rows, device, backend="host", halo=HALO):
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
        self.tiled = world > 1
        cls = HostBackend if backend == "host" else OracleBackend
        self.backend = cls(setup, self.window, self.tiled, device)
        self.frame = self.backend.frame
        self.device = self.backend.device
        self._xchg_s = 0.0
        self._gather_buf = None
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
        self.backend.run_pre()
        t0 = time.perf_counter()
        self.exchange_gather()
        self._xchg_s += time.perf_counter() - t0
        self.backend.run_main()
        self.backend.end_frame()
        t0 = time.perf_counter()
        self.exchange_history_halos()
        self._xchg_s += time.perf_counter() - t0

    def exchange_ms(self, steps):
        """host-side time spent issuing exchanges per step (device time shows in ms_per_step)"""
        return self._xchg_s / max(steps, 1) * 1e3

    # ---- exchange A: all-gather of the unbounded-reach surfaces -------------------------------------
    def _gather_plan(self):
        """[(src image, src mip, dst image, dst mip, divisor)]: tile interior at full-res >> divisor"""
        plan = [("depth", m, "frame_hiz", m - 1, m) for m in range(1, GATHER_MIPS + 1)]
        plan += [("dn", 0, "frame_normals", 0, 1), ("albedo", 0, "frame_albedo", 0, 0)]
        return plan

    def exchange_gather(self):
        plan = self._gather_plan()
        x0, y0, tw, th = self.tile
        sizes = []
        for src, mip, _, _, dv in plan:
            _, bpp, _ = self.backend.rows(src, mip)
            sizes.append((th >> dv) * (tw >> dv) * bpp)
        chunk = sum(sizes)
        if self._gather_buf is None:
            self._gather_buf = (torch.empty(chunk, dtype=torch.uint8, device=self.device),
                                torch.empty(chunk * self.world, dtype=torch.uint8, device=self.device))
        send, recv = self._gather_buf
        off = 0
        for (src, mip, _, _, dv), n in zip(plan, sizes):
            rows, bpp, (ox, oy, _, _) = self.backend.rows(src, mip)
            lx, ly, w, h = (x0 >> dv) - ox, (y0 >> dv) - oy, tw >> dv, th >> dv
            send[off: off + n].view(h, w * bpp).copy_(rows[ly: ly + h, lx * bpp: (lx + w) * bpp])
            off += n
        dist.all_gather_into_tensor(recv, send)
        for r in range(self.world):
            rx0, ry0, _, _ = tile_rect(r, self.cols, self.rows_n, self.tw, self.th)
            off = r * chunk
            for (_, _, dst, dmip, dv), n in zip(plan, sizes):
                rows, bpp, (ox, oy, _, _) = self.backend.rows(dst, dmip)
                lx, ly, w, h = (rx0 >> dv) - ox, (ry0 >> dv) - oy, tw >> dv, th >> dv
                rows[ly: ly + h, lx * bpp: (lx + w) * bpp].copy_(recv[off: off + n].view(h, w * bpp))
                off += n

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

    def exchange_history_halos(self):
        ops, unpack = [], []
        for name, dv in (("taa_hist", 0), ("acc_hist", 1), ("blurred_hist", 1)):
            rows, bpp, (ox, oy, ww, wh) = self.backend.rows(name)
            mine = tuple(v >> dv for v in self.tile)
            my_win = (ox, oy, ww, wh)
            for dx, dy, nb in self._neighbours():
                nb_tile = tuple(v >> dv for v in tile_rect(nb, self.cols, self.rows_n, self.tw, self.th))
                nb_win = tuple(v >> dv for v in window_rect(nb, self.cols, self.rows_n, self.tw, self.th, self.halo))
                # what the neighbour needs from me: my interior inside its window
                s = self._overlap(mine, nb_win)
                if s:
                    buf = rows[s[1] - oy: s[1] - oy + s[3], (s[0] - ox) * bpp: (s[0] - ox + s[2]) * bpp].contiguous()
                    ops.append(dist.P2POp(dist.isend, buf, nb))
                # what I need from it: its interior inside my window
                r = self._overlap(nb_tile, my_win)
                if r:
                    buf = torch.empty((r[3], r[2] * bpp), dtype=torch.uint8, device=self.device)
                    ops.append(dist.P2POp(dist.irecv, buf, nb))
                    unpack.append((rows, (r[0] - ox) * bpp, r[1] - oy, buf))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for rows, bx, ly, buf in unpack:
            rows[ly: ly + buf.shape[0], bx: bx + buf.shape[1]].copy_(buf)
