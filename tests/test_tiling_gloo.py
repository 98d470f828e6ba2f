"""Multi-process tiling rehearsed on the CPU: world_size 2 and 4 over gloo, compute
backend = oracle.  Every rank's tile interior must equal the single-process whole-frame result
bit for bit after three frames (all-gather of Hi-Z / normals / albedo + history halo exchange)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch.distributed as dist
    import vk_renderer_amd
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame, grid_for
    from vk_renderer_amd.chain import PostFxChain
    from oracle import binding
    binding.install()
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    cols, rows = (int(v) for v in os.environ['VKR_GRID'].split('x'))
    W, H = 128 * cols, int(os.environ.get('VKR_TILE_H', '144')) * rows
    setup = FrameSetup(W, H)
    mode = int(os.environ.get('VKR_GATHER_MODE', '1'))
    t = TiledFrame(setup, rank, world, cols, rows, None, backend=binding.OracleBackend, halo=48, gather_mode=mode)
    assert t.gather_mode == mode
    assert t.gather_mips == int(os.environ.get('VKR_EXPECT_GATHER', '4')), t.gather_mips
    t.prepare()
    for _ in range(3):  # the third frame reuses the exchange plans cached for the first (ping-pong parity)
        t.step()
    t.flush()
    ref = PostFxChain(W, H, backend='oracle', setup=setup)
    ref.synth(); ref.build_prev_hiz(); ref.init_histories(); ref.preintegrate_pdf()
    for _ in range(3):
        ref.frame(); ref.swap_histories()
    x0, y0, tw, th = t.tile
    c = t.backend.chain
    bad = 0
    for name, dv in (('rays', 1), ('raw', 1), ('reflections', 1), ('filtered', 1), ('blurred_hist', 1), ('acc_hist', 1), ('taa_hist', 0), ('dn', 1), ('dv', 1)):
        a, b = getattr(c, name), getattr(ref, name)
        ox, oy = a.origin
        sa = a.raw(0)[(y0 >> dv) - oy:(y0 >> dv) - oy + (th >> dv), (x0 >> dv) - ox:(x0 >> dv) - ox + (tw >> dv)]
        sb = b.raw(0)[(y0 >> dv):(y0 >> dv) + (th >> dv), (x0 >> dv):(x0 >> dv) + (tw >> dv)]
        n = int((sa.view(np.uint8) != sb.view(np.uint8)).sum())
        if n:
            print(f'rank {rank} {name}: {n} differing bytes')
        bad += n
    if mode != 1:  # hit colours (and, mode 0, hit normals) by request / reply: the exchange did carry something
        m = t.hit_matrix
        assert sum(sum(row) for row in m) > 0 and all(m[r][r] == 0 for r in range(world)), m
        if mode == 0:
            assert int(c.pend_mask.raw(0).astype(bool).sum()) >= 0
    # the gathered whole-frame pyramid equals the single-process pyramid (image mips 1..L-1)
    for m in range(c.frame_hiz.mips):
        n = int((c.frame_hiz.raw(m) != (ref.depth.raw(m + 1) & 0xFFFFFF)).sum())
        if n:
            print(f'rank {rank} frame_hiz mip {m}: {n} differing texels')
        bad += n
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# strips gather in place, 2-D grids pack and scatter; tile height 136 = 8 * 17 is divisible by 8 but not 16, like the
# 15360x1080 strips of BASELINE config 4 on 8 GPUs: only depth mips 1..3 are gathered, mip 4.. are rebuilt locally
@pytest.mark.parametrize("grid,tile_h,gather", [("1x2", 144, 4), ("2x1", 144, 4), ("2x2", 144, 4), ("1x2", 136, 3)])
def test_tiled_frame_matches_single_process(grid, tile_h, gather, tmp_path, oracle_lib):
    world = int(grid[0]) * int(grid[2])
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2", VKR_GRID=grid, VKR_TILE_H=str(tile_h), VKR_EXPECT_GATHER=str(gather))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("world,mode", [(2, 0), (4, 0), (2, 2)])
def test_request_reply_exchange_matches_single_process(world, mode, tmp_path, oracle_lib):
    """Hit colours and hit normals by request / reply (include/vkr_postfx.h vkr_hit_*, vkr_sssr_trace_windowed, vkr_sssr_validate)
    between real processes over gloo, on the oracle's twins of those entries: the albedo (mode 2) or the albedo and the
    downsampled normals (mode 0) are NOT gathered — every rank asks the owners for the footprint rows it lacks — and the tile
    interiors must still equal the single-process frame bit for bit after three frames."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2", VKR_GRID=f"1x{world}", VKR_TILE_H="144", VKR_EXPECT_GATHER="4", VKR_GATHER_MODE=str(mode))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_grid_and_windows():
    from vk_renderer_amd.tiling import grid_for

    assert grid_for(1) == (1, 1) and grid_for(2) == (1, 2) and grid_for(4) == (1, 4) and grid_for(8) == (1, 8)
    for cols, rows in (grid_for(8), (4, 2)):
        _check_grid(cols, rows)


def _check_grid(cols, rows):
    from vk_renderer_amd.tiling import tile_rect, window_rect

    seen = set()
    for r in range(8):
        x0, y0, w, h = tile_rect(r, cols, rows, 3840, 2160)
        assert (w, h) == (3840, 2160)
        seen.add((x0, y0))
        wx, wy, ww, wh = window_rect(r, cols, rows, 3840, 2160, 64)
        assert wx % 2 == 0 and wy % 2 == 0 and ww % 2 == 0 and wh % 2 == 0
        assert wx <= x0 and wy <= y0 and wx + ww >= x0 + w and wy + wh >= y0 + h
        assert 0 <= wx and wx + ww <= 3840 * cols and 0 <= wy and wy + wh <= 2160 * rows
    assert len(seen) == 8
