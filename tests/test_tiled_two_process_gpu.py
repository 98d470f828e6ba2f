"""Two real processes, one GPU: every rank is its own process with its own HostFrame on cuda:0, the exchanges go through
torch.distributed (gloo here — RCCL refuses two ranks on one device — so the bytes travel through host memory, but it
is the production TiledFrame.step(): coalesced all-gathers delivered in place, per-surface halo sends issued after the
producer pass and awaited before the consumer of the next frame).  Each rank checks its tile interior against the plain
single-GPU frame it computes itself."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import vk_renderer_amd
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    dist.init_process_group('gloo')
    cols, rows = (int(v) for v in os.environ['VKR_GRID'].split('x'))
    tw, th = 256, 160
    W, H = tw * cols, th * rows
    FRAMES = 3
    t = TiledFrame(FrameSetup(W, H), rank, world, cols, rows, device)
    t.prepare()
    for _ in range(FRAMES):
        t.step()
    t.flush()
    t.backend.sync()
    plain = TiledFrame(FrameSetup(W, H), 0, 1, 1, 1, device)
    plain.prepare()
    for _ in range(FRAMES):
        plain.step()
    plain.backend.sync()
    x0, y0, _, _ = t.tile
    bad = 0
    for name, dv in (('rays', 1), ('raw', 1), ('reflections', 1), ('filtered', 1), ('blurred_hist', 1), ('acc_hist', 1), ('taa_hist', 0), ('dn', 1), ('dv', 1)):
        got, want = t.frame.download(name), plain.frame.download(name)
        ox, oy = got.origin
        a = got.raw(0)[(y0 >> dv) - oy:(y0 >> dv) - oy + (th >> dv), (x0 >> dv) - ox:(x0 >> dv) - ox + (tw >> dv)]
        b = want.raw(0)[(y0 >> dv):(y0 >> dv) + (th >> dv), (x0 >> dv):(x0 >> dv) + (tw >> dv)]
        n = int((a != b).any(axis=-1).sum())
        if n:
            print(f'rank {rank} {name}: {n} differing texels')
        bad += n
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("grid", ["1x2", "2x1", "1x4"])
def test_ranks_as_processes_match_single_gpu_frame(grid, tmp_path):
    world = int(grid[0]) * int(grid[2])
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, VKR_GRID=grid, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
