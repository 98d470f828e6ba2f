"""Drop-in check, executed: the REFERENCE'S OWN pass sources driving the HIP kernels.

`make -C vk-renderer_amd/host refpasses` (run by __graft_entry__.build() wherever /root/reference is mounted) compiles
src/{downsample_pass,gtao,advanced_ssr,taa,defered_shading}.cpp of the reference, unchanged and where they lie, against the host mirror and
links them — instead of host/passes.cpp's own implementations of those five classes — with the mirror's rendergraph, gpu:: layer
and headless frame loop into host/build/refpasses/libvkr_host_refpasses.so.  Here one process loads that library, runs two
frames (reference code records every task: ten DownsampleDepth draws per frame, rand()-jittered GTAO angle pinned through
--wrap=rand, the raw Halton table ...) and compares every output with the oracle, exactly like the host-mirror parity test.
The built library travels to the GPU box; the reference sources do not have to.  Not an oracle: the arithmetic is the HIP
kernels'; what is under test is the boundary."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFLIB = os.path.join(ROOT, "vk-renderer_amd", "host", "build", "refpasses", "libvkr_host_refpasses.so")

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
    import numpy as np, torch
    import vk_renderer_amd
    from vk_renderer_amd import abi, host
    from vk_renderer_amd.camera import FrameSetup
    from vk_renderer_amd.chain import PostFxChain
    from oracle import binding
    from parity import mismatches
    binding.install()
    assert abi.HOST_LIB.endswith('libvkr_host_refpasses.so')
    W, H = (int(v) for v in sys.argv[1:3])
    setup = FrameSetup(W, H)
    frame = host.HostFrame(setup, device='cuda')
    frame.run(host.STAGE_LUT | host.STAGE_BRDF_LUT | host.STAGE_GBUFFER | host.STAGE_PREV_DEPTH)
    ref = PostFxChain(W, H, backend='oracle', setup=setup)
    ref.synth(); ref.build_prev_hiz(); ref.init_histories(); ref.preintegrate_pdf(); ref.preintegrate_brdf()
    frame.upload('taa_hist', ref.taa_hist.host)
    frame.upload('acc_hist', ref.acc_hist.host)
    angle_table = [60.0, 300.0, 180.0, 240.0, 120.0, 0.0, 300.0, 60.0, 180.0, 120.0, 240.0, 0.0]  # gtao.cpp:109
    for k in range(2):
        frame.run(host.STAGE_CHAIN | host.STAGE_SHADING)  # main.cpp:345-391 with the deferred composite between GTAO and TAA
        tasks = frame.last_tasks()
        frame.end_frame()
        ref.downsample(); ref.ssr_trace(frame_random=ref.frame_index %% 16); ref.ssr_filter(); ref.ssr_blur()
        ref.gtao_main(angle_offset=float(np.float32(angle_table[k %% 12]) / np.float32(360.0)))
        ref.gtao_filter(); ref.gtao_accumulate(); ref.shading(); ref.taa(color=ref.color_out)
        ref.frame_index += 1
        ref.swap_histories()
    torch.cuda.synchronize()
    # the reference records one DownsampleDepth task per mip (downsample_pass.cpp:107-129), the mirror's own pass one for the chain
    mips = ref.depth.mips
    want = ['DownsampleGbuffer'] + ['DownsampleDepth'] * (mips - 2) + ['SSSR_trace', 'SSSR_filter', 'SSSR_blur', 'GTAO_main', 'GTAO_filter',
                                                                   'GTAO_accumulate', 'DeferedShading', 'TAA']
    assert tasks == want, tasks
    bad_total = 0
    for hname, rimg, exact in (('color_out', ref.color_out, False), ('depth', ref.depth, True), ('dn', ref.dn, True), ('dv', ref.dv, True), ('rays', ref.rays, False),
                               ('reflections', ref.reflections, False), ('filtered', ref.filtered, False),
                               ('blurred_hist', ref.blurred_hist, False), ('acc_hist', ref.acc_hist, False), ('taa_hist', ref.taa_hist, False)):
        got = frame.download(hname)
        for mip in range(rimg.mips):
            if exact:
                a, b = got.raw(mip), rimg.raw(mip)
                if hname == 'depth':
                    a, b = a & 0xFFFFFF, b & 0xFFFFFF
                bad = int((a != b).any(axis=-1).sum())
                assert bad == 0, (hname, mip, bad)
            else:
                # one stored code is one storage step on sRGB8 surfaces too (tests/parity.py: storage_step), and the TAA history is a
                # blend of that sRGB8 composite (taa/resolve.comp): where the composite is one code apart (measured: 127 texels at
                # 1920x1080) the resolved colour may be one code gap apart.  Every surface has the same budget.
                bad = int(mismatches(rimg.format, got.decode(mip), rimg.decode(mip), input_fmt=abi.FMT_RGBA8_SRGB if hname == 'taa_hist' else None).sum())
                assert bad <= 8, (hname, bad)
            bad_total += bad
        print(f'[dropin] {hname:13s} outside-tol / differing {bad}')
    print(f'[dropin] reference pass sources over the mirror: {len(tasks)} tasks per frame, {bad_total} texels outside tolerance')
    frame.close()
""") % (ROOT, ROOT)


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="host/build/refpasses/libvkr_host_refpasses.so not built (needs the reference mounted at build time)")
@pytest.mark.parametrize("size", [(640, 360), (1920, 1080)])
def test_reference_pass_sources_drive_the_hip_kernels(size, tmp_path, oracle_lib):
    script = tmp_path / "dropin.py"
    script.write_text(WORKER)
    env = dict(os.environ, VKR_HOST_LIB=REFLIB)
    r = subprocess.run([sys.executable, str(script), str(size[0]), str(size[1])], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "[dropin] reference pass sources over the mirror" in r.stdout
