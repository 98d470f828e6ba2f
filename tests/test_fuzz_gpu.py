"""Seeded parameter fuzz of the hot path: random frame sizes (ragged against every tile size), cameras, push constants
and flag combinations, each pass fed the oracle's bytes and compared with the oracle's output under the rule of
tests/parity.py.  Catches what the fixed benchmark frame cannot: branches that only non-default parameters take
(non-MIS AO, two directions, reflections-only, cleared history, disabled blur / accumulation, every render_flags
combination, roughness cut-offs)."""
import os
import random

import numpy as np
import pytest

from vk_renderer_amd.camera import FrameSetup
from vk_renderer_amd.chain import PostFxChain

from parity import report
from test_parity_gpu import ALL_IMAGES, EXACT

pytestmark = pytest.mark.gpu

# VKR_FUZZ_SCALE=10 runs ten times as many seeds (a one-off campaign; the default keeps the suite short)
SCALE = int(os.environ.get("VKR_FUZZ_SCALE", "1"))


def _case(seed):
    rng = random.Random(0xC0FFEE + seed)
    w, h = 2 * rng.randint(33, 210), 2 * rng.randint(20, 130)
    setup = dict(frame_random=rng.randrange(16), use_mis=rng.choice((0, 1)),
                 eye=(rng.uniform(-1.5, 1.5), rng.uniform(0.4, 2.0), rng.uniform(-2.0, 0.0)), yaw=rng.uniform(60.0, 120.0),
                 prev_delta=(rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02), rng.uniform(-0.05, 0.05)),
                 prev_yaw_delta=rng.uniform(-0.6, 0.6))
    params = dict(
        ssr_trace=dict(max_roughness=rng.choice((1.0, 0.7, 0.35))),
        ssr_filter=dict(render_flags=rng.randrange(8)),
        ssr_blur=dict(max_roughness=rng.choice((1.0, 0.7, 0.35)), accumulate=rng.choice((0, 1)), disable_blur=rng.choice((0, 0, 1))),
        gtao_main=dict(angle_offset=rng.choice((0.0, 60.0, 120.0, 180.0, 240.0, 300.0)) / 360.0 + rng.uniform(-0.5, 0.5),
                       weight_ratio=rng.uniform(0.5, 2.0), two_directions=rng.choice((0, 255)), reflections_only=rng.choice((0, 0, 255))),
        gtao_accumulate=dict(clear_history=rng.choice((0, 0, 1))),
    )
    # (drawn last, so that the cases of earlier rounds keep their other values) half of the cases on the scene whose roughness
    # varies per texel: blur sigma differs from lane to lane, the resolve and GTAO see per-texel materials
    setup["material"] = rng.choice(("flat", "textured"))
    return w, h, setup, params


STAGES = [("downsample", ("depth", "dn", "dv")), ("ssr_trace", ("rays", "raw")), ("ssr_filter", ("reflections",)), ("ssr_blur", ("blurred",)),
          ("gtao_main", ("raw",)), ("gtao_filter", ("filtered",)), ("gtao_accumulate", ("acc_ao",)), ("taa", ("taa_target",))]


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_random_parameters_stagewise(seed, oracle_lib):
    import torch

    assert torch.cuda.is_available()
    w, h, setup_kw, params = _case(seed)
    print(f"[fuzz {seed}] {w}x{h} setup {setup_kw} params {params}")
    ref = PostFxChain(w, h, backend="oracle", setup=FrameSetup(w, h, **setup_kw))
    gpu = PostFxChain(w, h, backend="product", device="cuda", setup=FrameSetup(w, h, **setup_kw))
    ref.synth()
    ref.build_prev_hiz()
    ref.init_histories()
    ref.preintegrate_pdf()
    # a second oracle frame first, so the histories the passes read are not the seeded constants
    ref.frame()
    ref.swap_histories()
    total_bad = 0
    for stage, outs in STAGES:
        for name in ALL_IMAGES:
            getattr(gpu, name).copy_from(getattr(ref, name))
        kw = params.get(stage, {})
        getattr(ref, stage)(**kw)
        getattr(gpu, stage)(**kw)
        gpu.sync()
        for name in outs:
            r, g = getattr(ref, name), getattr(gpu, name)
            hg = g.to_host()
            for mip in range(r.mips):
                if name in EXACT:
                    a, b = g.raw(mip, hg), r.raw(mip)
                    if name == "depth":
                        a, b = a & 0xFFFFFF, b & 0xFFFFFF
                    assert not (a != b).any(), f"seed {seed} {stage} {name} mip {mip}: integer path must be bit-exact"
                else:
                    nbad, _ = report(f"{stage}:{name}.{mip}", r.format, g.decode(mip, hg), r.decode(mip))
                    total_bad += nbad
                    # a texel may flip a hit / break decision through libm-vs-ocml ulps in the smooth part
                    assert nbad <= max(1, int(2e-4 * r.width * r.height)), f"seed {seed} {stage} {name}: {nbad} texels outside tolerance"
    print(f"[fuzz {seed}] texels outside tolerance over all stages: {total_bad}")


def _prepared_pair(seed):
    rng = random.Random(0xBADC0DE + seed)
    w, h = 2 * rng.randint(40, 200), 2 * rng.randint(24, 120)
    setup_kw = dict(eye=(rng.uniform(-1.5, 1.5), rng.uniform(0.4, 2.0), rng.uniform(-2.0, 0.0)), yaw=rng.uniform(60.0, 120.0),
                    prev_delta=(rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02), rng.uniform(-0.05, 0.05)),
                    prev_yaw_delta=rng.uniform(-0.6, 0.6))
    ref = PostFxChain(w, h, backend="oracle", setup=FrameSetup(w, h, **setup_kw))
    gpu = PostFxChain(w, h, backend="product", device="cuda", setup=FrameSetup(w, h, **setup_kw))
    ref.synth()
    ref.build_prev_hiz()
    ref.init_histories()
    ref.preintegrate_pdf()
    ref.preintegrate_brdf()
    ref.frame()
    for name in ALL_IMAGES:
        getattr(gpu, name).copy_from(getattr(ref, name))
    gpu.preintegrate_brdf()
    gpu.brdf.copy_from(ref.brdf)
    print(f"[fuzz {seed}] {w}x{h} {setup_kw}")
    return rng, ref, gpu


def _check(seed, what, ref_img, gpu_img, gpu):
    gpu.sync()
    nbad, _ = report(what, ref_img.format, gpu_img.decode(), ref_img.decode())
    assert nbad <= max(1, int(2e-4 * ref_img.width * ref_img.height)), f"seed {seed} {what}: {nbad} texels outside tolerance"


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_random_parameters_widened_rows(seed, oracle_lib):
    """The passes outside the reference's frame loop and the rows either side of the hot path, same treatment: GTAO
    graphics / reprojection / deinterleaved, ScreenSpaceTrace, simple SSR, deferred shading, tile classification +
    indirect trace."""
    rng, ref, gpu = _prepared_pair(seed)
    # GTAO variants
    angle = rng.uniform(0.0, 1.0)
    ref.gtao_main_graphics(angle_offset=angle)
    gpu.gtao_main_graphics(angle_offset=angle)
    _check(seed, "gtao_graphics", ref.raw, gpu.raw, gpu)
    for c in (ref, gpu):
        c._layer_descs(2)
    ref.deinterleave_depth(2)
    gpu.deinterleave_depth(2)
    gpu.sync()
    for lr, lg in zip(ref.deint_layers, gpu.deint_layers):
        assert np.array_equal(lg.raw(0), lr.raw(0)), f"seed {seed}: deinterleaved depth layers must be bit-exact"
    layer = rng.randrange(16)
    ref.gtao_main_deinterleaved(layer=layer, angle_offset=angle)
    gpu.gtao_main_deinterleaved(layer=layer, angle_offset=angle)
    _check(seed, "gtao_deinterleaved", ref.raw, gpu.raw, gpu)
    # ScreenSpaceTrace
    st = dict(angle_offset=rng.uniform(0.0, 1.0), random_offset=rng.uniform(0.0, 1.0))
    ref.screen_trace(**st)
    gpu.screen_trace(**st)
    _check(seed, "screen_trace", ref.st_raw, gpu.st_raw, gpu)
    gpu.st_raw.copy_from(ref.st_raw)
    ref.screen_trace_filter()
    gpu.screen_trace_filter()
    _check(seed, "screen_trace_filter", ref.st_filtered, gpu.st_filtered, gpu)
    gpu.st_filtered.copy_from(ref.st_filtered)
    ref.screen_trace_accumulate()
    gpu.screen_trace_accumulate()
    _check(seed, "screen_trace_accumulate", ref.st_accumulated, gpu.st_accumulated, gpu)
    # simple SSR and the deferred-shading composite
    ref.ssr_simple()
    gpu.ssr_simple()
    _check(seed, "ssr_simple", ref.ssr_out, gpu.ssr_out, gpu)
    sh = dict(min_roughness=rng.uniform(0.0, 0.3), max_roughness=rng.uniform(0.5, 1.0), show_ao=rng.choice((0, 0, 1)))
    ref.shading(**sh)
    gpu.shading(**sh)
    _check(seed, "shading", ref.color_out, gpu.color_out, gpu)
    # tile classification (sets) + indirect trace
    cl = dict(max_roughness=rng.choice((1.0, 0.7, 0.35)), glossy_value=rng.uniform(0.2, 0.8))
    ref.ssr_classify(**cl)
    gpu.ssr_classify(**cl)
    gpu.sync()
    for name in ("reflective_tiles", "glossy_tiles"):
        nr = int(ref.buffer_to_host(getattr(ref, name.split("_")[0] + "_args"))[0])
        ng = int(gpu.buffer_to_host(getattr(gpu, name.split("_")[0] + "_args"))[0])
        assert nr == ng, f"seed {seed}: {name} count {ng} != {nr}"
        assert set(ref.buffer_to_host(getattr(ref, name))[:nr].tolist()) == set(gpu.buffer_to_host(getattr(gpu, name))[:ng].tolist())
    fr = rng.randrange(16)
    ref.ssr_trace_indirect(frame_random=fr, max_roughness=cl["max_roughness"])
    gpu.ssr_trace_indirect(frame_random=fr, max_roughness=cl["max_roughness"])
    _check(seed, "trace_indirect", ref.rays, gpu.rays, gpu)


@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_random_cameras_raster(seed, oracle_lib):
    """The raster stage under random cameras (triangles crossing the near plane, grazing views, sub-pixel and
    screen-filling triangles), sizes and tessellation: coverage and depth bit-exact, attachments within tolerance."""
    from vk_renderer_amd import scene as scn

    rng = random.Random(0xFACADE + seed)
    w, h = 2 * rng.randint(40, 260), 2 * rng.randint(24, 150)
    setup_kw = dict(eye=(rng.uniform(-3.0, 3.0), rng.uniform(0.15, 3.0), rng.uniform(-2.0, 5.0)), yaw=rng.uniform(30.0, 150.0),
                    prev_delta=(rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02), rng.uniform(-0.05, 0.05)),
                    prev_yaw_delta=rng.uniform(-0.6, 0.6))
    detail = rng.choice((6, 12, 24, 48))
    print(f"[fuzz {seed}] raster {w}x{h} detail {detail} {setup_kw}")
    sc = scn.procedural_scene(detail=detail)
    ref = PostFxChain(w, h, backend="oracle", setup=FrameSetup(w, h, **setup_kw))
    gpu = PostFxChain(w, h, backend="product", device="cuda", setup=FrameSetup(w, h, **setup_kw))
    for c in (ref, gpu):
        c.raster(sc)
    gpu.sync()
    a, b = gpu.depth.raw(0)[..., 0] & 0xFFFFFF, ref.depth.raw(0)[..., 0] & 0xFFFFFF
    assert np.array_equal(a, b), f"seed {seed}: {int((a != b).sum())} depth texels differ (coverage / depth must be bit-exact)"
    print(f"[fuzz {seed}] coverage {float((b != 0xFFFFFF).mean()):.3f}")
    for name in ("albedo", "normal", "material", "velocity"):
        r, g = getattr(ref, name), getattr(gpu, name)
        nbad, _ = report(name, r.format, g.decode(), r.decode())
        assert nbad <= max(1, int(2e-4 * w * h)), f"seed {seed} {name}: {nbad} texels outside tolerance"
