"""The C++ tiled frame (host/frame.cpp TiledFrame: strips, pack / unpack launches, exchanges issued from C++) on ONE GPU.

  * lockstep: every rank of a 1xN grid is an in-process C++ tiled frame without a communicator; the harness advances
    all ranks phase by phase (vkrh_tiled_phase) and plays the wire between phases by copying exactly the buffers the
    RCCL calls would move (vkrh_tiled_gather_parts / vkrh_tiled_halo_peers).  Tile interiors must equal the plain frame.
  * one rank through RCCL: a real one-rank communicator (vkr_comm_*: librccl loaded with dlopen), so vkr_all_gather and
    the event ordering between the compute and the exchange stream run for real.
The 15360x8640 / 8-strip frame of BASELINE config 4 is in tests/test_configs_gpu.py."""
import pytest

pytestmark = pytest.mark.gpu

OUTPUTS = (("rays", 1), ("raw", 1), ("reflections", 1), ("filtered", 1), ("blurred_hist", 1), ("acc_hist", 1), ("taa_hist", 0),
           ("dn", 1), ("dv", 1), ("depth", 0))


def bytes_at(ranks, addr, nbytes):
    """uint8 view of device memory [addr, addr + nbytes) owned by one of the ranks' allocators"""
    for t in ranks:
        try:
            tensor, off = t.frame.allocator.tensor_at(addr)
        except KeyError:
            continue
        assert off + nbytes <= tensor.numel()
        return tensor[off: off + nbytes]
    raise KeyError(hex(addr))


def move_gather(ranks, which):
    """what vkr_all_gather delivers: recv = [rank][bytes] of every rank's send"""
    parts = [t.frame.tiled_gather_parts(which) for t in ranks]
    for r, mine in enumerate(parts):
        for i, (_, recv, nbytes) in enumerate(mine):
            for src, theirs in enumerate(parts):
                send, _, n2 = theirs[i]
                assert n2 == nbytes
                bytes_at(ranks, recv + src * nbytes, nbytes).copy_(bytes_at(ranks, send, nbytes))


def move_halo(ranks, surface):
    """what vkr_halo_exchange delivers: every receive buffer gets the send buffer its peer packed for this rank"""
    peers = [t.frame.tiled_halo_peers(surface) for t in ranks]
    for r, mine in enumerate(peers):
        for peer, _, recv, nbytes in mine:
            send = [p for p in peers[peer] if p[0] == r][0][1]
            bytes_at(ranks, recv, nbytes).copy_(bytes_at(ranks, send, nbytes))


def lockstep_frame(ranks):
    for p in range(5):
        for t in ranks:
            t.frame.tiled_phase(p)
        if p == 0:      # both gathers start after the downsample; the harness completes them at once
            move_gather(ranks, 0)
            move_gather(ranks, 1)
        elif p == 1:
            move_halo(ranks, 0)
        elif p == 3:
            move_halo(ranks, 1)
        elif p == 4:
            move_halo(ranks, 2)
    for t in ranks:
        t._frame_no += 1


def _plain(W, H, frames, device):
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    plain = TiledFrame(FrameSetup(W, H), 0, 1, 1, 1, device)
    plain.prepare()
    for _ in range(frames):
        plain.step()
    plain.backend.sync()
    want = {n: plain.frame.download(n) for n, _ in OUTPUTS}
    plain.frame.close()
    return want


def _interiors_differ(t, want, tw, th):
    bad = 0
    x0, y0, _, _ = t.tile
    for name, dv in OUTPUTS:
        got = t.frame.download(name)
        ox, oy = got.origin
        a = got.raw(0)[(y0 >> dv) - oy:(y0 >> dv) - oy + (th >> dv), (x0 >> dv) - ox:(x0 >> dv) - ox + (tw >> dv)]
        b = want[name].raw(0)[(y0 >> dv):(y0 >> dv) + (th >> dv), (x0 >> dv):(x0 >> dv) + (tw >> dv)]
        if name == "depth":
            a, b = a & 0xFFFFFF, b & 0xFFFFFF
        n = int((a != b).any(axis=-1).sum())
        if n:
            print(f"rank {t.rank} {name}: {n} differing texels")
        bad += n
    return bad


@pytest.mark.parametrize("world,tile_h,gather", [(2, 160, 4), (4, 160, 4), (3, 136, 3)])  # 136 = 8 * 17: only depth mips 1..3 travel
def test_native_ranks_in_lockstep_match_single_gpu_frame(world, tile_h, gather):
    import torch

    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    tw, th = 256, tile_h
    W, H = tw, th * world
    device = torch.device("cuda", 0)
    want = _plain(W, H, 3, device)
    ranks = [TiledFrame(FrameSetup(W, H), r, world, 1, world, device, native=True, comm=None) for r in range(world)]
    for t in ranks:
        assert t.native and t.gather_mips == gather and t.frame.tiled_handle
        t.prepare()
    for _ in range(3):
        lockstep_frame(ranks)
    for t in ranks:
        t.flush()
    torch.cuda.synchronize()
    bad = sum(_interiors_differ(t, want, tw, th) for t in ranks)
    for t in ranks:
        t.frame.close()
    assert bad == 0


def test_native_tiled_frame_through_a_one_rank_rccl_communicator():
    """vkr_comm_unique_id / vkr_comm_create (librccl.so.1 via dlopen), grouped ncclAllGather launches on the frame's exchange
    stream, event ordering against the compute stream: three frames must equal the plain frame bit for bit."""
    import torch

    from vk_renderer_amd import abi
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    W, H = 512, 288
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    want = _plain(W, H, 3, device)
    comm = abi.Comm(0, 1, lambda ident: ident)
    t = TiledFrame(FrameSetup(W, H), 0, 1, 1, 1, device, force_tiled=True, native=True, comm=comm)
    assert t.tiled and t.native
    t.prepare()
    for _ in range(3):
        t.step()
    t.flush()
    torch.cuda.synchronize()
    bad = _interiors_differ(t, want, W, H)
    # the whole-frame pyramid the trace marched is the gathered one: mips 0..3 of frame_hiz == depth mips 1..4
    hiz, depth = t.frame.download("frame_hiz"), t.frame.download("depth")
    for m in range(4):
        assert (hiz.raw(m)[..., 0] & 0xFFFFFF == depth.raw(m + 1)[..., 0] & 0xFFFFFF).all()
    t.frame.close()
    comm.close()
    assert bad == 0
