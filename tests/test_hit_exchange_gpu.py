"""The multi-GPU entries of round 3, kernel by kernel against their oracle twins on one GPU: vkr_sssr_trace_windowed,
vkr_hit_requests (both passes), vkr_hit_reply, vkr_hit_scatter, vkr_sssr_validate (csrc/hit_exchange.hip, csrc/ssr.hip vs
oracle/passes_hit.cpp, oracle/passes_ssr.cpp).  One frame cut into two strips; the lower strip's window is traced with only
its own rows of the downsampled normals in memory, asks the upper strip for what it lacks, and must end with the rays the
plain trace stores.  Requests are an unordered set per owner on the GPU (block-aggregated atomics): compared sorted."""
import numpy as np
import pytest

from vk_renderer_amd import abi
from vk_renderer_amd.camera import FrameSetup
from vk_renderer_amd.chain import PostFxChain

from parity import mismatches

pytestmark = pytest.mark.gpu

W, H, HALO = 256, 320, 48
BOUNDS = [0, 160, 320]


def _window(rank):
    y0, y1 = BOUNDS[rank], BOUNDS[rank + 1]
    wy0, wy1 = max(0, y0 - HALO), min(H, y1 + HALO)
    return (0, wy0, W, wy1 - wy0)


def _chains(backend, device):
    setup = FrameSetup(W, H)
    plain = PostFxChain(W, H, backend=backend, device=device, setup=setup)
    ranks = [PostFxChain(W, H, backend=backend, device=device, setup=setup, window=_window(r), force_tiled=True) for r in range(2)]
    for c in [plain] + ranks:
        c.synth(); c.build_prev_hiz(); c.init_histories(); c.preintegrate_pdf(); c.downsample()
    plain.ssr_trace(frame_random=0)
    for c in ranks:  # the gathered pyramid: whole-frame image mips 1..L-1
        for m in range(c.frame_hiz.mips):
            src = plain.depth
            h = max(1, (H // 2) >> m)
            a = src.to_host()[src.offset[m + 1]: src.offset[m + 1] + src.pitch[m + 1] * h]
            dst = c.frame_hiz
            assert src.pitch[m + 1] == dst.pitch[m]
            host = dst.to_host().copy()
            host[dst.offset[m]: dst.offset[m] + dst.pitch[m] * h] = a
            dst.upload(host)
        c.hiz_tail(4)
        for src, dst in ((c.albedo, c.frame_albedo), (c.dn, c.frame_normals)):  # own rows only
            host = np.full(dst.nbytes, 0xA5, dtype=np.uint8)  # poison: a texel nobody delivers shows
            o = src.origin[1] * dst.pitch[0]
            host[o: o + src.height * src.pitch[0]] = src.to_host()[: src.height * src.pitch[0]]
            dst.upload(host)
    return plain, ranks


def test_windowed_trace_requests_replies_scatter_validate(oracle_lib):
    import torch

    assert torch.cuda.is_available()
    rp, rr = _chains("oracle", None)
    gp, gr = _chains("product", "cuda")
    lower_r, lower_g, upper_r, upper_g = rr[1], gr[1], rr[0], gr[0]
    # 1. windowed trace: rays (provisional), mask, pending data
    for c in (lower_r, lower_g):
        c.ssr_trace_windowed(frame_random=0)
    lower_g.sync()
    assert np.array_equal(lower_g.rays.raw(0), lower_r.rays.raw(0)), "windowed trace: rays differ"
    bad = int(mismatches(lower_r.raw.format, lower_g.raw.decode(0), lower_r.raw.decode(0)).sum())  # smooth terms: the usual tolerance
    assert bad <= 8, f"windowed trace: (occlusion, pdf): {bad} texels outside tolerance"
    mask_g, mask_r = lower_g.pend_mask.raw(0)[..., 0].astype(bool), lower_r.pend_mask.raw(0)[..., 0].astype(bool)
    assert np.array_equal(mask_g, mask_r) and mask_r.sum() > 0, f"pending masks differ / empty ({mask_r.sum()})"

    def pend(c):
        d = c.pend_data
        rows = d.to_host()[: d.pitch[0] * d.height].reshape(d.height, d.pitch[0])[:, : d.width * 16]
        return np.ascontiguousarray(rows).view(np.float32).reshape(d.height, d.width // 2, 8)

    pg, pr = pend(lower_g), pend(lower_r)
    assert np.array_equal(pg[mask_r][:, [0, 1, 2, 4, 5]], pr[mask_r][:, [0, 1, 2, 4, 5]]), "pending R / hit uv differ"
    # 2. requests: counts, then the set per owner
    cg, cr = lower_g.hit_count(BOUNDS), lower_r.hit_count(BOUNDS)
    assert cg == cr and cr[0] > 0 and cr[1] == 0, (cg, cr)
    qg, seg_g = lower_g.hit_write(BOUNDS, cg)
    qr, seg_r = lower_r.hit_write(BOUNDS, cr)
    qg_h, qr_h = lower_g.buffer_to_host(qg).view(np.uint32)[: seg_g[-1]], np.asarray(qr, dtype=np.uint32)[: seg_r[-1]]
    assert seg_g == seg_r and np.array_equal(np.sort(qg_h), np.sort(qr_h)), "request sets differ"
    assert (qr_h & abi.HIT_NORMAL).any() and (~qr_h & abi.HIT_NORMAL).any(), "both surfaces are asked for"
    assert (qr_h & abi.HIT_BOTH_ROWS).any()
    # 3. replies from the upper strip's window, on the same (sorted) request list
    order = np.sort(qr_h)
    n = int(order.size)
    rep_r, err_r = upper_r.hit_reply(order.copy(), n)
    q_dev = torch.from_numpy(order.view(np.int32).copy()).cuda()
    rep_g, err_g = upper_g.hit_reply(q_dev, n)
    assert err_r == 0 and err_g == 0
    rep_g_h, rep_r_h = upper_g.buffer_to_host(rep_g).view(np.uint32)[: 4 * n], np.asarray(rep_r, dtype=np.uint32)[: 4 * n]
    assert np.array_equal(rep_g_h, rep_r_h), "replies differ"
    # 4. scatter + deferred hit-normal test: the rays of the plain trace, and the requested texels in the frame images
    lower_r.hit_scatter(order.copy(), rep_r_h.copy(), n)
    lower_r.ssr_validate()
    lower_g.hit_scatter(q_dev, torch.from_numpy(rep_g_h.view(np.int32).copy()).cuda(), n)
    lower_g.ssr_validate()
    lower_g.sync()
    oy = lower_r.rays.origin[1]
    want = rp.rays.raw(0)[oy: oy + lower_r.rays.height]
    assert np.array_equal(lower_r.rays.raw(0), want), "oracle: validated rays differ from the plain trace"
    assert np.array_equal(lower_g.rays.raw(0), want), "product: validated rays differ from the plain trace"
    for name in ("frame_albedo", "frame_normals"):
        assert np.array_equal(getattr(lower_g, name).raw(0), getattr(lower_r, name).raw(0)), f"{name} differs after the scatter"
    # every texel the filter then reads was delivered: reflections equal the plain frame's on the strip's interior
    for c in (lower_r, lower_g, rp):
        c.ssr_filter()
    lower_g.sync()
    y0 = BOUNDS[1] // 2
    a = lower_g.reflections.raw(0)[y0 - oy:], lower_r.reflections.raw(0)[y0 - oy:]
    b = rp.reflections.raw(0)[y0:]
    assert np.array_equal(a[1], b) and np.array_equal(a[0], b), "reflections of the strip differ from the plain frame"


@pytest.mark.parametrize("rank,levels,park_after", [(1, 4, 2), (0, 4, 2), (1, 3, 0), (1, 1, 4), (0, 2, 1)])
def test_windowed_trace_local_rows_first(rank, levels, park_after, oracle_lib):
    """vkr_sssr_trace_windowed_head + _resume — the head marches on the rank's OWN rows of the first `levels` pyramid levels
    while the whole-frame pyramid is still poison, parks every ray at its first fetch of a texel that is not there, and the
    resume launch finishes them on the complete pyramid — against vkr_sssr_trace_windowed on the same strip: rays, (occlusion,
    pdf), pending mask and pending data bit for bit."""
    import torch

    _, gr = _chains("product", "cuda")
    c = gr[rank]
    c.ssr_trace_windowed(frame_random=0)
    c.sync()
    mask, data = c._pending_images()
    want = {n: getattr(c, n).raw(0).copy() for n in ("rays", "raw")}
    want_mask, want_data = mask.raw(0).copy(), data.to_host().copy()
    pend = want_mask[..., 0].astype(bool)
    for img in (c.rays, c.raw, mask, data):  # whatever the two launches forget to write shows
        img.upload(np.full(img.nbytes, 0x5A, dtype=np.uint8))
    hiz_bytes = c.frame_hiz.to_host().copy()
    c.frame_hiz.upload(np.full(c.frame_hiz.nbytes, 0xEE, dtype=np.uint8))  # "not arrived yet": the head must not read a byte of it
    c.ssr_trace_windowed_head(levels, frame_random=0, park_after=park_after)
    c.sync()
    parked = int(c._trace_workspace[:4].view(torch.int32)[0].item())
    total = c.rays.width * c.rays.height
    print(f"[local-first] rank {rank}, {levels} local levels, park after {park_after} rounds: {parked} of {total} rays parked ({parked / total:.3f})")
    assert 0 < parked < total
    c.frame_hiz.upload(hiz_bytes)
    c.ssr_trace_windowed_resume(frame_random=0)
    c.sync()
    for n in ("rays", "raw"):
        assert np.array_equal(getattr(c, n).raw(0), want[n]), f"{n} differs from the one-launch windowed trace"
    assert np.array_equal(mask.raw(0), want_mask), "pending mask differs"
    d = data.to_host().reshape(-1)
    rows_g = d[: data.pitch[0] * data.height].reshape(data.height, data.pitch[0])[:, : data.width * 16].view(np.float32).reshape(data.height, data.width // 2, 8)
    rows_w = want_data.reshape(-1)[: data.pitch[0] * data.height].reshape(data.height, data.pitch[0])[:, : data.width * 16].view(np.float32).reshape(data.height, data.width // 2, 8)
    assert np.array_equal(rows_g[pend][:, [0, 1, 2, 4, 5]], rows_w[pend][:, [0, 1, 2, 4, 5]]), "pending R / hit uv differ"


def test_requests_into_segments_of_fixed_room(oracle_lib):
    """vkr_hit_requests_bounded (the native wire sizes its messages from the PREVIOUS frame's counts): with room for every request
    the segments hold exactly the exact pass's sets and nothing is flagged; with too little room every segment is full, holds a
    subset of the exact set, the overflow is flagged, and the unused slots of a roomy segment say VKR_HIT_NO_REQUEST — which the
    reply answers with zeros without counting an error."""
    _, gr = _chains("product", "cuda")
    lower, upper = gr[1], gr[0]
    lower.ssr_trace_windowed(frame_random=0)
    counts = lower.hit_count(BOUNDS)
    exact, seg = lower.hit_write(BOUNDS, counts)
    exact_h = lower.buffer_to_host(exact).view(np.uint32)[: seg[-1]]
    n0 = counts[0]
    assert n0 > 200 and counts[1] == 0
    # roomy: 100 slots more than needed for owner 0, 64 for owner 1 (which gets nothing)
    out, s2, dropped = lower.hit_write_bounded(BOUNDS, [n0 + 100, 64])
    h = lower.buffer_to_host(out).view(np.uint32)
    assert dropped == 0
    assert np.array_equal(np.sort(h[: n0]), np.sort(exact_h[: n0])), "a roomy segment holds the exact set"
    assert (h[n0: s2[-1]] == 0xFFFFFFFF).all(), "unused slots say: no request"
    rep, err = upper.hit_reply(out, s2[1])
    rep_h = upper.buffer_to_host(rep).view(np.uint32)[: 4 * s2[1]].reshape(-1, 4)
    assert err == 0 and (rep_h[n0:] == 0).all() and rep_h[:n0].any()
    # tight: half the room
    cap = n0 // 2
    out, s3, dropped = lower.hit_write_bounded(BOUNDS, [cap, 64])
    h = lower.buffer_to_host(out).view(np.uint32)
    assert dropped == 1
    kept = h[:cap]
    assert (kept != 0xFFFFFFFF).all() and np.isin(kept, exact_h[: n0]).all() and (h[cap: s3[-1]] == 0xFFFFFFFF).all()
