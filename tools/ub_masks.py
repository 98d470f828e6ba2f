#!/usr/bin/env python3
"""How many output texels depend on behaviour the reference leaves undefined?

Runs the ORACLE (never the product) on a BASELINE configuration twice:
  mode 0  the frozen rule: texelFetch outside the frame / beyond the view's last mip returns 0;
  mode 1  the other plausible hardware behaviour: clamp to the edge texel / the last mip;
and reports, per pass, (a) the output pixels whose evaluation performed at least one such fetch (counted inside the
oracle, formats.hpp UbPixel) and (b) the output texels whose stored value differs between the two runs — the texels
a real Vulkan run could legitimately disagree on.  For the storage-qualifier mismatches (A.5: `r8` / `rg8` / `rgba8`
declared, R16F / RG16F / RGBA16F bound) every texel of the image goes through the frozen rule "the host-created format
wins"; the report gives the largest change the other reading (quantise to 8 bits) would make.

    python tools/ub_masks.py --config c2        -> profiles/ub_mask_c2.json
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from oracle import binding  # noqa: E402
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.chain import PostFxChain  # noqa: E402

PASSES = ["SSSR_trace", "SSSR_filter", "SSSR_blur", "GTAO_main", "GTAO_filter", "GTAO_accumulate", "TAA"]
SIZES = {"c1": (1920, 1080), "c2": (3840, 2160), "c5": (3840, 2160)}


def run(lib, cfg, mode):
    W, H = SIZES[cfg]
    lib.vkr_ref_ub_set_oob_mode(mode)
    lib.vkr_ref_ub_reset()
    c = PostFxChain(W, H, backend="oracle", setup=FrameSetup(W, H, use_mis=0 if cfg == "c1" else 1))
    c.synth(); c.build_prev_hiz(); c.init_histories(); c.preintegrate_pdf()
    lib.vkr_ref_ub_reset()  # prev-Hi-Z / LUT construction is not part of the frame
    if cfg == "c1":
        c.downsample(); c.gtao_main()
        outs = ("raw",)
    elif cfg == "c2":
        c.frame()
        outs = ("rays", "raw", "reflections", "blurred", "filtered", "acc_ao", "taa_target")
    else:
        c.downsample()
        for k in range(8):
            c.ssr_trace(frame_random=k); c.ssr_filter(); c.ssr_blur()
        c.taa()
        outs = ("rays", "raw", "reflections", "blurred", "taa_target")
    counts = (C.c_uint64 * (len(PASSES) * 2))()
    lib.vkr_ref_ub_counts(counts)
    lib.vkr_ref_ub_set_oob_mode(0)
    return c, outs, [(int(counts[2 * i]), int(counts[2 * i + 1])) for i in range(len(PASSES))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=sorted(SIZES))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = binding.install()
    lib.vkr_ref_ub_counts.argtypes = [C.c_void_p]
    W, H = SIZES[a.config]
    frozen, outs, counts = run(lib, a.config, 0)
    alt, _, _ = run(lib, a.config, 1)
    launches = 8 if a.config == "c5" else 1
    report = {
        "config": a.config, "frame": [W, H], "checker": "oracle (CPU restatement), never the product",
        "rule_frozen": "texelFetch outside the frame or beyond the view's last mip returns 0 (SURVEY Appendix A.4)",
        "rule_alternative": "clamp to the edge texel / the last mip",
        "pixels_that_performed_an_undefined_fetch": {
            p: {"out_of_frame": oof, "beyond_last_mip": blm, "pass_executions": launches if p.startswith("SSSR") else 1}
            for p, (oof, blm) in zip(PASSES, counts) if oof or blm},
        "texels_whose_value_changes_under_the_alternative": {},
        "format_qualifier_rule": {},
    }
    for name in outs:
        f, g = getattr(frozen, name), getattr(alt, name)
        n = int((f.raw(0) != g.raw(0)).any(axis=-1).sum())
        d = np.nan_to_num(np.abs(f.decode().astype(np.float64) - g.decode().astype(np.float64)))
        report["texels_whose_value_changes_under_the_alternative"][name] = {
            "texels": f.width * f.height, "changed": n, "fraction": n / (f.width * f.height), "max_abs_change": float(d.max())}
    # A.5: declared 8-bit UNORM qualifier vs the 16-bit float image actually bound
    for name, decl, shader in (("filtered", "r8", "gtao/filter.comp:8"), ("acc_ao", "rg8", "gtao/accum.comp:10"), ("taa_target", "rgba8", "taa/resolve.comp:9")):
        if name not in outs:
            continue
        v = np.nan_to_num(getattr(frozen, name).decode().astype(np.float64))
        q = np.rint(np.clip(v, 0.0, 1.0) * 255.0) / 255.0
        report["format_qualifier_rule"][name] = {
            "declared": decl, "shader": shader, "texels_governed_by_the_rule": int(v.shape[0] * v.shape[1]),
            "max_abs_change_if_the_qualifier_won": float(np.abs(v - q).max()), "values_outside_0_1": int(((v < 0) | (v > 1)).any(axis=-1).sum())}
    out = a.out or os.path.join(ROOT, "profiles", f"ub_mask_{a.config}.json")
    with open(out, "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
