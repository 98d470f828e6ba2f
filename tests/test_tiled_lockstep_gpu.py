"""The multi-GPU configuration on ONE GPU: every rank of a 2x2 (and a 2x1) grid is an in-process TiledFrame over its own
window (tile + halo, origin != 0) driven through the C++ host mirror and the HIP kernels; the exchanges are played by
copying the packed buffers between the in-process ranks (same pack / unpack code as the RCCL path, only the wire is
replaced).  After two frames every rank's tile interior must equal the plain single-GPU frame bit for bit — this is
the windowed addressing of every kernel, the whole-frame Hi-Z / normals / albedo path and the history halos, on the
product."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OUTPUTS = (("rays", 1), ("raw", 1), ("reflections", 1), ("filtered", 1), ("blurred_hist", 1), ("acc_hist", 1), ("taa_hist", 0),
           ("dn", 1), ("dv", 1), ("depth", 0))


def _lockstep_frame(ranks):
    world = len(ranks)
    for t in ranks:
        t.backend.run_stage("downsample")
    for group, stages_before, stages_after in (("hiz", ("taa",), ("trace", "gtao")), ("albedo", (), ("ssr_resolve",))):
        packed = [t.gather_pack(group) for t in ranks]
        for t in ranks:
            for st in stages_before:
                t.backend.run_stage(st)
        for r, t in enumerate(ranks):
            send, recv, plan, sizes, chunk = packed[r]
            for src in range(world):  # what all_gather_into_tensor delivers
                recv[src * chunk: (src + 1) * chunk].copy_(packed[src][0])
            t.gather_unpack(plan, sizes, chunk, recv)
            for st in stages_after:
                t.backend.run_stage(st)
    for t in ranks:
        t.backend.end_frame()
    plans = [t.halo_pack() for t in ranks]
    for r, plan in enumerate(plans):
        for nb, send, sbuf, recv, rbuf in plan:
            if recv:  # the matching send buffer of neighbour `nb` towards rank r
                peer = [p for p in plans[nb] if p[0] == r][0]
                assert peer[2].numel() == rbuf.numel()
                rbuf.copy_(peer[2])
    for t, plan in zip(ranks, plans):
        t.halo_unpack(plan)


@pytest.mark.parametrize("grid", [(2, 1), (2, 2)])
def test_tiled_ranks_match_single_gpu_frame(grid):
    import torch

    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    cols, rows = grid
    world = cols * rows
    tw, th = 256, 160  # divisible by 16
    W, H = tw * cols, th * rows
    device = torch.device("cuda", 0)

    plain = TiledFrame(FrameSetup(W, H), 0, 1, 1, 1, device)
    plain.prepare()
    for _ in range(2):
        plain.step()
    plain.backend.sync()
    want = {n: plain.frame.download(n) for n, _ in OUTPUTS}
    plain.frame.close()

    ranks = [TiledFrame(FrameSetup(W, H), r, world, cols, rows, device, halo=64) for r in range(world)]  # production halo: covers the 19 half-res px reach of GTAO main + filter
    for t in ranks:
        assert t.tiled and t.window != (0, 0, W, H)
        t.prepare()
    for _ in range(2):
        _lockstep_frame(ranks)
    torch.cuda.synchronize()
    bad = 0
    for r, t in enumerate(ranks):
        x0, y0, _, _ = t.tile
        for name, dv in OUTPUTS:
            got = t.frame.download(name)
            ox, oy = got.origin
            a = got.raw(0)[(y0 >> dv) - (oy >> 0):(y0 >> dv) - oy + (th >> dv), (x0 >> dv) - ox:(x0 >> dv) - ox + (tw >> dv)]
            b = want[name].raw(0)[(y0 >> dv):(y0 >> dv) + (th >> dv), (x0 >> dv):(x0 >> dv) + (tw >> dv)]
            if name == "depth":
                a, b = a & 0xFFFFFF, b & 0xFFFFFF
            n = int((a != b).any(axis=-1).sum())
            if n:
                print(f"rank {r} {name}: {n} differing texels")
            bad += n
        t.frame.close()
    assert bad == 0
