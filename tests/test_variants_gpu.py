"""GPU parity for the passes the reference ships but its frame loop never records (SURVEY.md 8(a) rows
G4 and R2): graphics GTAO, static AO reprojection, deinterleaved GTAO, ScreenSpaceTrace — HIP vs the
CPU oracle on identical synthetic inputs, through the C-ABI."""
import numpy as np
import pytest

from vk_renderer_amd import abi

from parity import report
from test_parity_gpu import _compare, _pair, _sync_inputs

pytestmark = pytest.mark.gpu


def _prepared(size, oracle_lib):
    ref, gpu = _pair(*size, oracle_lib)
    ref.synth()
    ref.build_prev_hiz()
    ref.downsample()
    _sync_inputs(ref, gpu)
    return ref, gpu


def _cmp_extra(ref, gpu, name):
    gpu.sync()
    r, g = getattr(ref, name), getattr(gpu, name)
    n, _ = report(name, r.format, g.decode(), r.decode())
    assert n <= 1e-4 * r.width * r.height, f"{name}: {n} texels outside tolerance"
    return r


@pytest.mark.parametrize("size", [(256, 144), (640, 360)])
def test_gtao_graphics_variant(size, oracle_lib):
    """gtao/main.frag (program "gtao_main"): 20 samples, radius min(200/|P|, 32), sky -> 1."""
    ref, gpu = _prepared(size, oracle_lib)
    for angle in (60.0 / 360.0, 300.0 / 360.0 + 0.137):
        ref.gtao_main_graphics(angle_offset=angle)
        gpu.gtao_main_graphics(angle_offset=angle)
        _compare(ref, gpu, ("raw",), budget=1e-4)
    ao = ref.raw.decode()[..., 0]
    sky = ref.depth.decode(1)[..., 0] >= 1.0
    assert sky.any() and np.all(ao[sky] == 1.0), "sky pixels of the graphics variant must store 1"
    assert 0.05 < float(ao[~sky].mean()) < 1.5


def test_gtao_reproject_variant(oracle_lib):
    """gtao/reproject.comp, STATIC_REPROJECT: blend only where depth is unchanged to 1e-6."""
    ref, gpu = _prepared((256, 144), oracle_lib)
    ref.setup.use_mis = 0
    ref.gtao_main()
    ref.gtao_filter()
    # half of the frame keeps its depth (prev := cur), the other half keeps the moved-camera prev depth
    cur, prev = ref.depth.to_host().copy(), ref.prev_depth.to_host().copy()
    prev[: len(prev) // 2] = cur[: len(cur) // 2]
    ref.prev_depth.upload(prev)
    rng = np.random.default_rng(5)
    for c in (ref, gpu):
        c._half_img(abi.FMT_R16_SFLOAT, "ao_prev_frame")
        c._half_img(abi.FMT_R16_SFLOAT, "ao_output")
    hist = ref.ao_prev_frame.to_host()
    h16 = hist.view(np.uint16)
    h16[:] = rng.integers(0, 0x3C00, size=h16.shape, dtype=np.uint16)  # fp16 codes in [0, 1)
    ref.ao_prev_frame.upload(hist)
    _sync_inputs(ref, gpu)
    gpu.ao_prev_frame.copy_from(ref.ao_prev_frame)
    ref.gtao_reproject()
    gpu.gtao_reproject()
    r = _cmp_extra(ref, gpu, "ao_output")
    blended = int((r.decode()[..., 0] != ref.filtered.decode()[..., 0]).sum())
    print(f"[parity] reproject blended texels {blended}")
    assert blended > 0.2 * r.width * r.height and blended < 0.8 * r.width * r.height


def test_gtao_deinterleaved_variant(oracle_lib):
    """gtao_opt/{deinterleave,main_deinterleaved}.comp with the reference's literal dispatch sizes."""
    ref, gpu = _prepared((512, 288), oracle_lib)
    for c in (ref, gpu):
        c._layer_descs(2)
    ref.deinterleave_depth(2)
    gpu.deinterleave_depth(2)
    gpu.sync()
    touched = 0
    for lr, lg in zip(ref.deint_layers, gpu.deint_layers):
        a, b = lg.raw(0), lr.raw(0)
        assert np.array_equal(a, b), "deinterleaved depth layers must be bit-exact"
        touched += int((b != 0).sum())
    assert touched > 0
    for layer in (0, 5, 15):
        ref.gtao_main_deinterleaved(layer=layer)
        gpu.gtao_main_deinterleaved(layer=layer)
        _compare(ref, gpu, ("raw",), budget=1e-4)


@pytest.mark.parametrize("size", [(256, 144), (640, 360)])
def test_screen_space_trace(size, oracle_lib):
    """screen_trace/{trace,filter,accumulate}.comp (ScreenSpaceTrace, row R2)."""
    ref, gpu = _prepared(size, oracle_lib)
    ref.screen_trace()
    gpu.screen_trace()
    r = _cmp_extra(ref, gpu, "st_raw")
    raw = r.decode()
    lit = int((raw[..., :3].max(axis=-1) > 0).sum())
    print(f"[parity] screen_trace lit texels {lit}, mean ao term {float(raw[..., 3].mean()):.4f}")
    assert lit > 0.005 * r.width * r.height, "the test scene does not exercise the hit path of ScreenSpaceTrace"
    gpu.st_raw.copy_from(ref.st_raw)
    ref.screen_trace_filter()
    gpu.screen_trace_filter()
    _cmp_extra(ref, gpu, "st_filtered")
    gpu.st_filtered.copy_from(ref.st_filtered)
    # accumulate twice: first against an empty history, then in place against itself; make depth static on half
    cur, prev = ref.depth.to_host().copy(), ref.prev_depth.to_host().copy()
    prev[: len(prev) // 2] = cur[: len(cur) // 2]
    ref.prev_depth.upload(prev)
    gpu.prev_depth.copy_from(ref.prev_depth)
    for _ in range(2):
        ref.screen_trace_accumulate()
        gpu.screen_trace_accumulate()
        _cmp_extra(ref, gpu, "st_accumulated")


def test_host_mirror_variants(oracle_lib):
    """GTAO::add_main_pass_graphics / add_reprojection_pass / deinterleave_depth / add_main_pass_deinterleaved and
    ScreenSpaceTrace recorded through the C++ rendergraph mirror produce what the oracle computes on the same inputs."""
    from vk_renderer_amd import host
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.chain import PostFxChain
    from parity import mismatches

    W, H = 512, 288
    setup = FrameSetup(W, H)
    frame = host.HostFrame(setup, device="cuda")
    frame.run(host.STAGE_GBUFFER | host.STAGE_PREV_DEPTH)
    frame.run(host.STAGE_DOWNSAMPLE)
    ref = PostFxChain(W, H, backend="oracle", setup=setup)
    for name in ("depth", "prev_depth", "normal", "albedo", "material", "velocity"):
        getattr(ref, name).upload(frame.download(name).to_host())

    def check(host_name, ref_img, layer=None):
        got = frame.download(host_name, layer)
        bad = int(mismatches(ref_img.format, got.decode(), ref_img.decode()).sum())
        print(f"[parity] host {host_name:20s} outside-tol {bad}")
        assert bad <= 1e-4 * ref_img.width * ref_img.height, f"{host_name}: {bad} texels differ"

    # graphics GTAO -> filter -> static reprojection (frame_count 0 => 60 deg, jitter pinned to 0)
    frame.pin_randoms(0.0, 0, 0)
    frame.run(host.STAGE_GTAO_GRAPHICS)
    assert frame.last_tasks() == ["GTAO", "GTAO_filter", "GTAO_reproject"]
    ref.gtao_main_graphics(angle_offset=60.0 / 360.0)
    check("raw", ref.raw)
    ref.gtao_filter()
    check("filtered", ref.filtered)
    ref.gtao_reproject()
    check("ao_output", ref.ao_output)
    # deinterleaved: frame_count is now 1 => 300 deg
    frame.run(host.STAGE_GTAO_DEINTERLEAVED)
    assert frame.last_tasks() == ["GTAO_deinterleave", "GTAO_deinterleaved"]
    ref.deinterleave_depth(2)
    for layer in (0, 7, 15):
        check("deinterleaved_depth", ref.deint_layers[layer], layer)
    ref.gtao_main_deinterleaved(layer=0, angle_offset=300.0 / 360.0)
    check("raw", ref.raw)
    # ScreenSpaceTrace
    frame.pin_screen_trace(0.0, 0.25, 0)
    frame.run(host.STAGE_SCREEN_TRACE)
    assert frame.last_tasks() == ["ScreenTrace", "ScreenTraceFilter", "ScreenTraceAccumulate"]
    ref.screen_trace(angle_offset=60.0 / 360.0, random_offset=0.25)
    check("st_raw", ref.st_raw)
    ref.screen_trace_filter()
    check("st_filtered", ref.st_filtered)
    ref.screen_trace_accumulate()
    check("st_accumulated", ref.st_accumulated)
    frame.close()


def test_host_mirror_reprojection_two_frames(oracle_lib):
    """main.cpp:417 remaps gtao.output <-> gtao.prev_frame at the end of every frame: the static-reprojection variant
    (gtao.cpp:241-284) must read in frame 2 what it wrote in frame 1."""
    from vk_renderer_amd import host
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.chain import PostFxChain
    from parity import mismatches

    W, H = 512, 288
    setup = FrameSetup(W, H)
    frame = host.HostFrame(setup, device="cuda")
    frame.run(host.STAGE_GBUFFER | host.STAGE_PREV_DEPTH)
    frame.run(host.STAGE_DOWNSAMPLE)
    ref = PostFxChain(W, H, backend="oracle", setup=setup)
    for name in ("depth", "prev_depth", "normal", "albedo", "material", "velocity"):
        getattr(ref, name).upload(frame.download(name).to_host())
    frame.pin_randoms(0.0, 0, 0)
    for k, angle in enumerate((60.0, 300.0)):  # gtao.cpp:109: table[frame_count % 12], jitter pinned to 0
        frame.run(host.STAGE_GTAO_GRAPHICS)
        ref.gtao_main_graphics(angle_offset=float(np.float32(angle) / np.float32(360.0)))
        ref.gtao_filter()
        ref.gtao_reproject()
        got = frame.download("ao_output")
        bad = int(mismatches(ref.ao_output.format, got.decode(), ref.ao_output.decode()).sum())
        print(f"[parity] reprojection frame {k}: ao_output outside-tol {bad}")
        assert bad <= 1e-4 * ref.ao_output.width * ref.ao_output.height
        if k == 1:  # the history actually contributed: frame 2 differs from a run without it
            assert not np.array_equal(got.raw(0), first), "frame 2 ignored the history"
        first = got.raw(0).copy()
        frame.end_frame()
        ref.ao_output, ref.ao_prev_frame = ref.ao_prev_frame, ref.ao_output
    frame.close()


def test_readback_captures(tmp_path, oracle_lib):
    """ReadBackSystem (image_readback.cpp) + capture writers on real device images: the CSV / PNG files hold exactly
    the bytes of the images they were read from."""
    import numpy as np
    from PIL import Image

    from vk_renderer_amd import host
    from vk_renderer_amd.camera import FrameSetup

    W, H = 256, 144
    frame = host.HostFrame(FrameSetup(W, H), device="cuda")
    frame.run(host.STAGE_GBUFFER | host.STAGE_PREV_DEPTH)
    frame.run(host.STAGE_DOWNSAMPLE)
    for mip in (0, 1):
        frame.capture("depth", tmp_path / f"d{mip}.csv", host.HostFrame.CAPTURE_DEPTH_CSV, mip=mip)
        want = frame.download("depth").raw(mip)[..., 0] & 0xFFFFFF
        rows = (tmp_path / f"d{mip}.csv").read_text().split("\n")[1:-1]
        got = np.array([[int(c, 16) for c in r.split(",")[1:]] for r in rows], dtype=np.uint32)
        assert np.array_equal(got, want)
    frame.capture("depth", tmp_path / "d.png", host.HostFrame.CAPTURE_DEPTH_PNG)
    png = np.array(Image.open(tmp_path / "d.png")).astype(np.uint32)
    want = frame.download("depth").raw(0)[..., 0] & 0xFFFFFF
    assert np.array_equal(png[..., 0] | (png[..., 1] << 8) | (png[..., 2] << 16), want) and np.all(png[..., 3] == 0)
    frame.capture("albedo", tmp_path / "a.png", host.HostFrame.CAPTURE_RGBA_PNG)
    col = np.array(Image.open(tmp_path / "a.png"))
    raw = frame.download("albedo").raw(0)
    assert np.array_equal(col[..., :3], raw[..., :3]) and np.all(col[..., 3] == 255)
    frame.close()


@pytest.mark.parametrize("size,glossy", [((256, 144), 0.5), ((648, 360), 0.35)])
def test_tile_classified_trace(size, glossy, oracle_lib):
    """SSSR_Clear + classification.comp + trace_indirect.comp.  The tile lists are sets (the shader
    appends with atomics); the rays image they produce is compared texel by texel."""
    ref, gpu = _prepared(size, oracle_lib)
    for c in (ref, gpu):
        c.ssr_classify(glossy_value=glossy)
    ra, ga = ref.buffer_to_host(ref.reflective_args), gpu.buffer_to_host(gpu.reflective_args)
    rg, gg = ref.buffer_to_host(ref.glossy_args), gpu.buffer_to_host(gpu.glossy_args)
    assert list(ra) == list(ga) and list(rg) == list(gg) and list(ra[1:]) == [1, 1]
    w2, h2 = size[0] // 2, size[1] // 2
    assert int(ra[0]) + int(rg[0]) == ((w2 + 7) // 8) * ((h2 + 7) // 8)
    print(f"[parity] tiles: reflective {int(ra[0])} glossy {int(rg[0])}")
    assert int(ra[0]) > 0 and int(rg[0]) > 0, "the test scene must populate both tile classes"
    for name, n in (("reflective_tiles", int(ra[0])), ("glossy_tiles", int(rg[0]))):
        a = np.sort(ref.buffer_to_host(getattr(ref, name))[:n])
        b = np.sort(gpu.buffer_to_host(getattr(gpu, name))[:n])
        assert np.array_equal(a, b), f"{name}: tile sets differ"
    # sentinel fill: every texel of the window must be written by exactly one of the two dispatches
    for c in (ref, gpu):
        h = c.rays.to_host()
        h[:] = 0x5A
        c.rays.upload(h)
        c.ssr_trace_indirect(frame_random=3)
    _compare(ref, gpu, ("rays",), budget=1e-4)
    raw = ref.rays.raw(0)
    assert not np.all(raw == 0x5A5A, axis=-1).any(), "some texels were not traced"
    valid = int((raw[..., 3] != 0xFFFF).sum())
    print(f"[parity] indirect trace valid hits {valid}")
    assert valid > 0.01 * w2 * h2


def test_host_mirror_classified_ssr(oracle_lib):
    """AdvancedSSR::run with Settings::use_tile_classification: SSSR_Clear, SSSR_Classification, two indirect trace
    dispatches, then filter and blur — through the rendergraph mirror, against the oracle chain."""
    from vk_renderer_amd import host
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.chain import PostFxChain
    from parity import mismatches

    W, H = 512, 288
    setup = FrameSetup(W, H)
    frame = host.HostFrame(setup, device="cuda")
    frame.run(host.STAGE_LUT | host.STAGE_GBUFFER | host.STAGE_PREV_DEPTH)
    frame.run(host.STAGE_DOWNSAMPLE)
    ref = PostFxChain(W, H, backend="oracle", setup=setup)
    for name in ("depth", "prev_depth", "normal", "albedo", "material", "velocity", "dn", "dv"):
        getattr(ref, name).upload(frame.download(name).to_host())
    frame.pin_randoms(0.0, 0, 5)
    frame.run(host.STAGE_SSR_CLASSIFIED)
    assert frame.last_tasks() == ["SSSR_Clear", "SSSR_Classification", "SSSR_trace", "SSSR_filter", "SSSR_blur"]
    ref.ssr_classify()
    nr, ng = int(ref.reflective_args[0]), int(ref.glossy_args[0])
    assert list(frame.read_buffer("reflective_indirect")) == [nr, 1, 1] and list(frame.read_buffer("glossy_indirect")) == [ng, 1, 1]
    assert np.array_equal(np.sort(frame.read_buffer("reflective_tiles")[:nr]), np.sort(ref.reflective_tiles[:nr]))
    assert np.array_equal(np.sort(frame.read_buffer("glossy_tiles")[:ng]), np.sort(ref.glossy_tiles[:ng]))
    ref.ssr_trace_indirect(frame_random=5)
    ref.ssr_filter()
    ref.ssr_blur()
    for host_name, img in (("rays", ref.rays), ("reflections", ref.reflections), ("blurred", ref.blurred)):
        bad = int(mismatches(img.format, frame.download(host_name).decode(), img.decode()).sum())
        print(f"[parity] host {host_name:14s} outside-tol {bad}")
        assert bad <= 1e-4 * img.width * img.height
    frame.close()
