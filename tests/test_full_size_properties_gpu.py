"""BASELINE-size (3840x2160) checks of the HIP path through size-independent properties (the oracle takes seconds per
frame at this size; parity against it runs at smaller sizes in test_parity_gpu.py):
  - Hi-Z: every mip texel is the minimum of its 2x2 parents, odd extents drop the last row / column;
  - the downsampled normal / velocity of a half-res pixel are those of the full-res texel that holds the minimum depth;
  - the chain is deterministic: two runs from the same state are bit-identical (the trace compacts rays with atomics,
    which must not leak into results);
  - TAA: zero velocity and history == colour  =>  0.9 clamp(colour, neighbour box) + 0.1 colour; constant reflections stay
    constant under the blur."""
import numpy as np
import pytest

from vk_renderer_amd import abi
from vk_renderer_amd.chain import PostFxChain

pytestmark = pytest.mark.gpu
W, H = 3840, 2160


@pytest.fixture(scope="module")
def chain():
    c = PostFxChain(W, H, backend="product", device="cuda")
    c.synth()
    c.build_prev_hiz()
    c.init_histories()
    c.preintegrate_pdf()
    c.downsample()
    c.sync()
    return c


def test_hiz_min_property_4k(chain):
    host = chain.depth.to_host()
    prev = chain.depth.raw(0, host)[..., 0] & 0xFFFFFF
    for m in range(1, chain.depth.mips):
        cur = chain.depth.raw(m, host)[..., 0] & 0xFFFFFF
        h, w = cur.shape
        p = prev[: 2 * h, : 2 * w]  # odd parent extents: last row / column dropped (depth_mips.frag:7-15)
        want = np.minimum(np.minimum(p[0::2, 0::2], p[0::2, 1::2]), np.minimum(p[1::2, 0::2], p[1::2, 1::2]))
        assert np.array_equal(cur, want), f"mip {m}"
        prev = cur
    assert chain.depth.mips == 12 and prev.shape == (1, 1)


def test_downsample_selects_the_min_depth_texel_4k(chain):
    d0 = chain.depth.raw(0)[..., 0] & 0xFFFFFF
    n0, v0 = chain.normal.raw(0), chain.velocity.raw(0)
    dn, dv = chain.dn.raw(0), chain.dv.raw(0)
    quad = np.stack([d0[0::2, 0::2], d0[0::2, 1::2], d0[1::2, 0::2], d0[1::2, 1::2]])  # d0 d1 d2 d3 (downsample_gbuffer.frag:14-17)
    mn = quad.min(axis=0)
    # offset of the first of d1, d2, d3 equal to the minimum, else (0, 0) (:21-31)
    pick = np.where(quad[1] == mn, 1, np.where(quad[2] == mn, 2, np.where(quad[3] == mn, 3, 0)))
    oy, ox = pick >> 1, pick & 1
    yy, xx = np.mgrid[0:H // 2, 0:W // 2]
    assert np.array_equal(dn, n0[2 * yy + oy, 2 * xx + ox])
    assert np.array_equal(dv, v0[2 * yy + oy, 2 * xx + ox])
    assert np.array_equal(chain.depth.raw(1)[..., 0] & 0xFFFFFF, mn)


def test_frame_is_deterministic_4k(chain):
    import torch

    names = ("rays", "raw", "reflections", "blurred", "filtered", "acc_ao", "taa_target")
    runs = []
    for _ in range(2):
        chain.init_histories()
        chain.frame_index = 0
        chain.frame()
        chain.sync()
        runs.append({n: getattr(chain, n).to_host().copy() for n in names})
    for n in names:
        assert np.array_equal(runs[0][n], runs[1][n]), f"{n}: two runs from the same state differ"
    torch.cuda.synchronize()


def test_taa_identity_and_blur_constant_4k(chain):
    from vk_renderer_amd.images import ImageBuf

    # TAA: velocity 0, history = colour (init_histories) -> mix(clamp(hist), cur, 0.1) == cur up to fp16 storage
    chain.init_histories()
    keep = chain.velocity
    chain.velocity = ImageBuf(abi.FMT_RG16_SFLOAT, W, H, device="cuda")
    chain.taa()
    chain.velocity = keep
    chain.sync()
    out = chain.taa_target.decode()[..., :3]
    col = chain.albedo.decode()[..., :3]
    hist = col.astype(np.float16).astype(np.float32)  # what init_histories stored
    p = np.pad(col, ((1, 1), (1, 1), (0, 0)), mode="edge")
    nb = np.stack([p[1:-1, :-2], p[1:-1, 2:], p[:-2, 1:-1], p[2:, 1:-1]])  # the four textureOffset neighbours (resolve.comp:37-46)
    want = 0.9 * np.clip(hist, nb.min(axis=0), nb.max(axis=0)) + 0.1 * col
    # the shader samples bilinearly at texel centres: weights are 0 up to the rounding of (x + 0.5) / W * W - 0.5
    assert np.abs(out - want).max() <= 2.0 ** -8, "TAA with zero velocity: mix(clamp(history, neighbour box), colour, 0.1)"
    # blur: constant reflections (and no history) stay constant wherever any tap carries weight
    h = chain.reflections.to_host()
    h[:] = 0
    h.view(np.uint8).reshape(-1, 4)[:, :3] = 100
    chain.reflections.upload(h)
    hh = chain.blurred_hist.to_host()
    hh[:] = 0
    hh.view(np.uint8).reshape(-1, 4)[:, :3] = 100
    chain.blurred_hist.upload(hh)
    chain.ssr_blur()
    chain.sync()
    b = chain.blurred.raw(0)[..., :3]
    frac_const = float((b == 100).all(axis=-1).mean())
    assert frac_const > 0.999, f"only {frac_const:.4f} of the blurred texels kept the constant"
