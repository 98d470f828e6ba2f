#!/usr/bin/env python3
"""Generates tests/golden/chain_256x144.npz from the CPU oracle (the reference ships no golden
vectors and cannot run here, SURVEY.md 8(c), so these are self-generated pins: they freeze the
oracle's behaviour, they do not prove it equals the reference's).

Content: for the frozen frame setup (camera.FrameSetup defaults, SEED 0x5EED0001) at 256x144,
two frames of the chain with history ping-pong; per image a SHA-256 of the raw storage bytes of
every mip and a 24x16 crop of raw storage values around the image centre.

One pair of fixtures per numeric contract (oracle/glsl.hpp VKR_CONTRACT): chain_256x144_contract<N>.npz and
extras_256x144_contract<N>.npz, N = what the oracle library on disk was built with (`make -C oracle CONTRACT=1|2`).

Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vk_renderer_amd  # noqa: E402,F401
from vk_renderer_amd.chain import PostFxChain  # noqa: E402

W, H = 256, 144
IMAGES = ("depth", "prev_depth", "normal", "albedo", "material", "velocity", "dn", "dv", "rays", "raw", "reflections", "filtered",
          "blurred_hist", "acc_hist", "taa_hist")


def contract():
    """numeric contract of the oracle library on disk"""
    from oracle import binding

    lib = binding.install(build_if_missing=True)
    return int(lib.vkr_ref_numeric_contract())


def fixture(stem):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{stem}_contract{contract()}.npz")


def _ensure_backend(backend):
    if backend == "oracle":
        from oracle import binding

        binding.install(build_if_missing=True)


def run_chain(backend="oracle", device=None, material="flat"):
    _ensure_backend(backend)
    from vk_renderer_amd.camera import FrameSetup

    c = PostFxChain(W, H, backend=backend, device=device, setup=FrameSetup(W, H, material=material))
    c.synth()
    c.build_prev_hiz()
    c.init_histories()
    c.preintegrate_pdf()
    for _ in range(2):
        c.frame()
        c.swap_histories()
    c.sync()
    return c


# second fixture: the rows added around the chain (8(f) #1, #4)
EXTRA_IMAGES = ("ssr_out", "color_out", "brdf_crop", "raw_graphics", "ao_output", "raw_deinterleaved", "st_raw", "st_filtered",
                "st_accumulated", "rays_indirect")


def run_extras(backend="oracle", device=None):
    """{name: ImageBuf} after running each widened pass once on the frozen 256x144 scene."""
    from vk_renderer_amd.images import ImageBuf

    _ensure_backend(backend)
    c = run_chain(backend, device)
    out = {}

    def snap(name, img):
        b = ImageBuf(img.format, img.width, img.height, img.mips)
        b.upload(img.to_host())
        out[name] = b

    c.ssr_simple()
    snap("ssr_out", c.ssr_out)
    c.preintegrate_brdf()
    c.shading()
    snap("color_out", c.color_out)
    crop_src = c.brdf.raw(0)[::64, ::64].copy()  # 16 x 16 spot grid of the 1024^2 split-sum LUT
    lut = ImageBuf(c.brdf.format, crop_src.shape[1], crop_src.shape[0])
    lut.set_raw(crop_src)
    out["brdf_crop"] = lut
    c.gtao_main_graphics()
    snap("raw_graphics", c.raw)
    c.gtao_filter()
    c.gtao_reproject()
    snap("ao_output", c.ao_output)
    c.deinterleave_depth(2)
    c.gtao_main_deinterleaved(layer=5)
    snap("raw_deinterleaved", c.raw)
    c.screen_trace()
    snap("st_raw", c.st_raw)
    c.screen_trace_filter()
    snap("st_filtered", c.st_filtered)
    c.screen_trace_accumulate()
    snap("st_accumulated", c.st_accumulated)
    c.ssr_classify()
    c.ssr_trace_indirect(frame_random=3)
    snap("rays_indirect", c.rays)
    c.sync()
    return out


def digest(img):
    host = img.to_host()
    out = []
    for m in range(img.mips):
        raw = np.ascontiguousarray(img.raw(m, host))
        if img.format == 1:  # D24: stencil bits are not part of the contract
            raw = raw & 0xFFFFFF
        out.append(hashlib.sha256(raw.tobytes()).hexdigest())
    return out


def crop(img):
    raw = img.raw(0)
    cy, cx = raw.shape[0] // 2, raw.shape[1] // 2
    return np.ascontiguousarray(raw[cy - 8:cy + 8, cx - 12:cx + 12])


def main():
    c = run_chain()
    data = {}
    for name in IMAGES:
        img = getattr(c, name)
        data[name + "__sha256"] = np.array(digest(img))
        data[name + "__crop"] = crop(img)
    data["pdf__spot"] = c.pdf.decode()[[100, 512, 900], :, 0][:, [100, 512, 900]]
    out = fixture("chain_256x144")
    np.savez_compressed(out, **data)
    print("wrote", out, os.path.getsize(out), "bytes")

    # the second material mode (VKR_SYNTH_TEXTURED_ROUGHNESS): roughness per texel
    c = run_chain(material="textured")
    data = {}
    for name in IMAGES:
        img = getattr(c, name)
        data[name + "__sha256"] = np.array(digest(img))
        data[name + "__crop"] = crop(img)
    out = fixture("textured_256x144")
    np.savez_compressed(out, **data)
    print("wrote", out, os.path.getsize(out), "bytes")

    extras = run_extras()
    data = {}
    for name in EXTRA_IMAGES:
        data[name + "__sha256"] = np.array(digest(extras[name]))
        data[name + "__crop"] = crop(extras[name]) if extras[name].height >= 16 and extras[name].width >= 24 else extras[name].raw(0)
    out = fixture("extras_256x144")
    np.savez_compressed(out, **data)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
