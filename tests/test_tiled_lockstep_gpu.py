"""The multi-GPU configuration on ONE GPU: every rank of a grid (2-D: 2x1, 2x2; strips: 1x2, 1x4) is an in-process TiledFrame over its own
window (tile + halo, origin != 0) driven through the C++ host mirror and the HIP kernels; the exchanges are played by
copying the packed buffers between the in-process ranks (same pack / unpack code as the RCCL path, only the wire is
replaced).  After three frames every rank's tile interior must equal the plain single-GPU frame bit for bit — this is
the windowed addressing of every kernel, the whole-frame Hi-Z / normals / albedo path and the history halos, on the
product."""
import pytest

pytestmark = pytest.mark.gpu

OUTPUTS = (("rays", 1), ("raw", 1), ("reflections", 1), ("filtered", 1), ("blurred_hist", 1), ("acc_hist", 1), ("taa_hist", 0),
           ("dn", 1), ("dv", 1), ("depth", 0))


def _move_halos(ranks, which):
    """what the point-to-point sends deliver: every receive buffer gets the matching send buffer of its neighbour"""
    for r, t in enumerate(ranks):
        for nb, _, rbuf in t.halo_peers(which):
            if rbuf is not None:
                sbuf = [p for p in ranks[nb].halo_peers(which) if p[0] == r][0][1]
                assert sbuf.numel() == rbuf.numel()
                rbuf.copy_(sbuf)


def _lockstep_frame(ranks):
    """Advances every rank's TiledFrame.phases() — the production frame order — one exchange point at a time and
    plays the wire in between."""
    world = len(ranks)
    gens = [t.phases() for t in ranks]
    while True:
        ops = [next(g, None) for g in gens]
        if ops[0] is None:
            assert all(o is None for o in ops)
            return
        kind = ops[0][0]
        assert all(o[0] == kind for o in ops), "ranks diverged"
        if kind == "gather_wait":  # what all_gather_into_tensor delivers
            for _, g in ops:
                for i, (_, recv) in enumerate(g.parts):
                    for src in range(world):
                        recv.view(world, -1)[src].copy_(ops[src][1].parts[i][0])
        elif kind == "halo_wait":
            _move_halos(ranks, ops[0][1])


@pytest.mark.parametrize("grid", [(2, 1), (2, 2), (1, 2), (1, 4)])  # 2-D grids: packed gathers; strips: in place
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
    for _ in range(3):
        plain.step()
    plain.backend.sync()
    want = {n: plain.frame.download(n) for n, _ in OUTPUTS}
    plain.frame.close()

    ranks = [TiledFrame(FrameSetup(W, H), r, world, cols, rows, device, halo=48) for r in range(world)]  # production halo: covers the 20 half-res px reach of GTAO main + filter
    for t in ranks:
        assert t.tiled and t.window != (0, 0, W, H)
        t.prepare()
    for _ in range(3):  # the third frame reuses the exchange plans cached for the first (ping-pong parity)
        _lockstep_frame(ranks)
    for which in ("taa", "ao", "ssr"):  # the refreshes the last frame left in flight
        _move_halos(ranks, which)
    for t in ranks:
        t.flush()
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
