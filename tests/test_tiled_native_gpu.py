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


from vk_renderer_amd.tiling import native_lockstep_frame as lockstep_frame  # noqa: E402  (the harness: plays the wire with copies)


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


@pytest.mark.parametrize("world,tile_h,gather,mode,local_first,whole", [(2, 160, 4, 0, False, False), (4, 160, 4, 0, False, False), (3, 136, 3, 0, False, False),
                                                                          (4, 160, 4, 1, False, False), (4, 160, 4, 2, False, False), (4, 160, 4, 0, True, False),
                                                                          (3, 136, 3, 0, True, False),  # 136 = 8 * 17: only depth mips 1..3 travel
                                                                          (4, 160, 4, 0, False, True), (3, 136, 3, 0, True, True)])
def test_native_ranks_in_lockstep_match_single_gpu_frame(world, tile_h, gather, mode, local_first, whole, monkeypatch):
    """mode 0 (the default): hit colours AND hit normals by request / reply (vkr_sssr_trace_windowed, vkr_hit_requests /
    _reply / _scatter, vkr_sssr_validate); 1: the albedo and the downsampled normals of the whole frame all-gathered into
    every rank (round 2); 2: albedo by request, normals gathered.  All must equal the plain frame on every tile interior.
    local_first (VKR_TILED_LOCAL_FIRST=1, off by default): the trace in two stages around the depth gather — head on the
    rank's own pyramid rows, resume on the whole-frame pyramid — and the TAA behind GTAO.
    whole (VKR_TILED_WHOLE_WINDOW=1): every pass on its whole window, as in round 3; by default a rank computes only the rows
    of its window that something reads (host/frame.cpp: clip_outputs)."""
    import torch

    monkeypatch.setenv("VKR_TILED_WHOLE_WINDOW", "1" if whole else "0")
    monkeypatch.setenv("VKR_TILED_GATHER_MODE", str(mode))
    monkeypatch.setenv("VKR_TILED_LOCAL_FIRST", "1" if local_first else "0")
    by_gather = mode == 1

    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    tw, th = 256, tile_h
    W, H = tw, th * world
    device = torch.device("cuda", 0)
    want = _plain(W, H, 3, device)
    ranks = [TiledFrame(FrameSetup(W, H), r, world, 1, world, device, native=True, comm=None) for r in range(world)]
    for t in ranks:
        assert t.native and t.gather_mips == gather and t.frame.tiled_handle
        assert t.frame.tiled_local_first() == local_first
        t.prepare()
    for _ in range(3):
        lockstep_frame(ranks)
    for t in ranks:
        t.flush()
    torch.cuda.synchronize()
    bad = sum(_interiors_differ(t, want, tw, th) for t in ranks)
    if not by_gather:
        m = ranks[0].hit_matrix  # [requester][owner] of the last frame
        assert sum(m) > 0 and all(m[r * world + r] == 0 for r in range(world)), f"hit footprints cross the strips of this frame: {m}"
        assert all(t.frame.tiled_hit_errors() == 0 for t in ranks)
        # far fewer bytes than the albedo of the other strips
        assert 20 * sum(m) < 4 * W * H * (world - 1) // world
    # the gathered group: depth mips only when the hit normals travel by request
    assert all(len(t.frame.tiled_gather_parts(0)) == gather + (0 if mode == 0 else 1) for t in ranks)
    if mode == 0:
        pending = sum(int(t.frame.download("pend_mask").raw(0).astype(bool).sum()) for t in ranks)
        assert pending > 0, "no ray of this frame ends on another rank's rows: the deferred hit-normal test was never exercised"
        print(f"[tiled] pending hit-normal tests in the last frame: {pending}")
    for t in ranks:
        t.frame.close()
    assert bad == 0


@pytest.mark.parametrize("bounds,gather", [([0, 160, 400, 480], 4), ([0, 96, 168, 304, 480], 3)])  # 168 = 8 * 21: mips 1..3 travel
def test_strips_of_different_heights_match_single_gpu_frame(bounds, gather):
    """Cost-balanced strips (vkrh_tiled_config.row_bounds): every rank owns a different number of rows; the shares of the
    whole-frame surfaces then lie at different offsets (what vkr_all_gather_v moves).  Interiors must equal the plain frame."""
    import torch

    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    world, W, H = len(bounds) - 1, 256, bounds[-1]
    device = torch.device("cuda", 0)
    want = _plain(W, H, 3, device)
    ranks = [TiledFrame(FrameSetup(W, H), r, world, 1, world, device, native=True, comm=None, row_bounds=bounds) for r in range(world)]
    for r, t in enumerate(ranks):
        assert t.native and t.gather_mips == gather and t.tile == (0, bounds[r], W, bounds[r + 1] - bounds[r])
        t.prepare()
    for _ in range(3):
        lockstep_frame(ranks)
    for t in ranks:
        t.flush()
    torch.cuda.synchronize()
    bad = sum(_interiors_differ(t, want, W, t.th) for t in ranks)
    for t in ranks:
        t.frame.close()
    assert bad == 0


@pytest.mark.parametrize("gather_v", [False, True])
def test_native_tiled_frame_through_a_one_rank_rccl_communicator(gather_v, monkeypatch):
    """vkr_comm_unique_id / vkr_comm_create (librccl.so.1 via dlopen), grouped ncclAllGather launches on the frame's exchange
    stream, event ordering against the compute stream: three frames must equal the plain frame bit for bit.
    gather_v: the same through vkr_all_gather_v (grouped ncclBroadcast launches: what strips of different heights use)."""
    import torch

    if gather_v:
        monkeypatch.setenv("VKR_TILED_FORCE_GATHER_V", "1")
    else:
        monkeypatch.delenv("VKR_TILED_FORCE_GATHER_V", raising=False)

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


def test_one_rank_without_force_tiled_has_no_phases():
    """A one-rank frame that is not forced onto the multi-GPU path has no exchange stream and no events: vkrh_tiled_step runs
    the plain chain, vkrh_tiled_phase refuses with a message.  The Python harness does not even make a native frame for it
    (TiledFrame.native is False, its handle None): the vkrh_tiled_* entries must answer a NULL handle with an error, not a
    crash (tools/lockstep_profile.py --world 1 ran into exactly that)."""
    import torch

    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import HostBackend, TiledFrame

    W, H = 256, 128
    device = torch.device("cuda", 0)
    t = TiledFrame(FrameSetup(W, H), 0, 1, 1, 1, device, native=True, comm=None)
    assert not t.tiled and not t.native and t.frame.tiled_handle is None
    for call in (lambda: t.frame.tiled_phase(0), t.frame.tiled_step, t.frame.tiled_flush, lambda: t.frame.tiled_time_waits(True)):
        with pytest.raises(RuntimeError, match="NULL tiled frame"):
            call()
    t.frame.close()
    # the C++ tiled frame itself, one rank, not forced: step is the plain chain, phases do not exist
    b = HostBackend(FrameSetup(W, H), (0, 0, W, H), False, device,
                    native=dict(rank=0, world=1, halo=48, gathered_mips=4, force_tiled=False, comm=None, row_bounds=None))
    b.prepare()
    with pytest.raises(RuntimeError, match="not tiled"):
        b.frame.tiled_phase(0)
    for _ in range(2):
        b.frame.tiled_step()
    b.frame.tiled_flush()
    torch.cuda.synchronize()
    assert b.frame.download("taa_hist").raw(0).any()  # the chain ran
    b.frame.close()


@pytest.mark.parametrize("pipelined", [False, True])
def test_native_frame_on_an_emulated_wire_receives_what_its_peers_would_send(pipelined, monkeypatch):
    """tools/wire_emulation.py in small: three ranks are driven in lockstep with real data (the SSR frame counter pinned, so every
    frame asks for the same hit texels; the last frame's hit segments in the native layout, host.hit_capacities), then every rank
    goes on NATIVELY — its own exchange stream, events, the hit round enqueued on the previous frame's counts — on an emulated
    communicator that moves nothing and holds the stream for the wire's time (vkr_comm_create_emulated).  What the passes without
    a history produce (rays, raw, reflections, filtered) must still be the plain frame's, bit for bit: the stale receive buffers
    are what the peers would send — vkr_hit_requests hands out the slots of a segment in a fixed order, so this frame's requests
    sit where the replies of the last lockstep frame answer them —, no request names a texel its owner does not hold, every
    round goes out on the seeded capacities, and nothing runs ahead of the exchange it needs.
    pipelined (VKR_TILED_PIPELINE=1): two frames in flight — the next frame's downsample (into the G-buffer's second set) and depth
    all-gather start right after this frame's trace, the TAA runs behind GTAO (host/frame.cpp: pipelined_step); same images."""
    import torch

    monkeypatch.setenv("VKR_TILED_PIPELINE", "1" if pipelined else "0")

    from vk_renderer_amd import abi
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.tiling import TiledFrame

    world, tw, th = 3, 256, 160
    W, H = tw, th * world
    device = torch.device("cuda", 0)
    plain = TiledFrame(FrameSetup(W, H), 0, 1, 1, 1, device)
    plain.prepare()
    for _ in range(3):
        plain.frame.pin_randoms(0.0, 0, 0)
        plain.step()
    plain.backend.sync()
    names = ("rays", "raw", "reflections", "filtered")
    want = {n: plain.frame.download(n) for n in names}
    plain.frame.close()

    ranks = [TiledFrame(FrameSetup(W, H), r, world, 1, world, device, native=True, comm=None) for r in range(world)]
    for t in ranks:
        t.prepare()
    for k in range(3):
        for t in ranks:
            t.frame.pin_randoms(0.0, 0, 0)
        lockstep_frame(ranks, hit_in_capacities=(k == 2))
    counts = ranks[0].hit_matrix
    assert sum(counts) > 0
    bad = 0
    for r, t in enumerate(ranks):
        comm = abi.Comm.emulated(r, world, 60.0, 5.0)
        assert t.frame.tiled_pipelined() == pipelined
        t.frame.tiled_emulate_wire(comm.handle, counts)
        for _ in range(3):
            t.frame.pin_randoms(0.0, 0, 0)
            t.frame.tiled_step()
        t.frame.tiled_flush()
        torch.cuda.synchronize()
        assert t.frame.tiled_hit_errors() == 0
        assert t.frame.tiled_hit_rounds() == (3, 0, 0), "every round must go out on the seeded capacities, none repeated"
        x0, y0, _, _ = t.tile
        for name in names:
            got = t.frame.download(name)
            ox, oy = got.origin
            a = got.raw(0)[(y0 >> 1) - oy:(y0 >> 1) - oy + (th >> 1), (x0 >> 1) - ox:(x0 >> 1) - ox + (tw >> 1)]
            b = want[name].raw(0)[(y0 >> 1):(y0 >> 1) + (th >> 1), (x0 >> 1):(x0 >> 1) + (tw >> 1)]
            n = int((a != b).any(axis=-1).sum())
            if n:
                print(f"rank {r} {name}: {n} differing texels")
            bad += n
        t.frame.close()
        comm.close()
    assert bad == 0
