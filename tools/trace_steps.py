#!/usr/bin/env python3
"""Per-ray step counts of the SSR march on a BASELINE frame, from the ORACLE (test infrastructure, never the product):
writes an (h2, w2) uint8 .npy for tools/trace_sim.py.

    python tools/trace_steps.py --size 3840 2160 --out /tmp/sim/steps_4k.npy
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import vk_renderer_amd  # noqa: E402,F401
from oracle import binding  # noqa: E402
from vk_renderer_amd.camera import FrameSetup  # noqa: E402
from vk_renderer_amd.chain import PostFxChain  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=(3840, 2160))
    ap.add_argument("--out", default="/tmp/sim/steps.npy")
    a = ap.parse_args()
    lib = binding.install()
    W, H = a.size
    c = PostFxChain(W, H, backend="oracle", setup=FrameSetup(W, H))
    c.synth(); c.build_prev_hiz(); c.init_histories(); c.preintegrate_pdf()
    c.downsample()
    steps = np.zeros((H // 2, W // 2), dtype=np.uint8)
    lib.vkr_ref_set_step_sink.argtypes = [C.c_void_p, C.c_int]
    lib.vkr_ref_set_step_sink(steps.ctypes.data, steps.strides[0])
    gate = np.zeros(8, dtype=np.uint64)
    lib.vkr_ref_set_gate_counters.argtypes = [C.c_void_p]
    lib.vkr_ref_set_gate_counters(gate.ctypes.data)
    c.ssr_trace(frame_random=0)
    lib.vkr_ref_set_step_sink(None, 0)
    lib.vkr_ref_set_gate_counters(None)
    for name, k in (("pinned steps", gate[:4]), ("later steps", gate[4:])):
        print(f"{name}: {int(k[0])} steps, horizon update evaluated (mip <= 1) {int(k[1])}, gate passed {int(k[2])}, |v.z| >= 0.3 alone refuses {int(k[3])}")
    np.save(a.out, steps)
    h = np.bincount(steps.ravel(), minlength=81)
    print("mean", steps.mean(), "hist16", h[16] / steps.size, "hist80", h[80] / steps.size, "min", steps.min())


if __name__ == "__main__":
    main()
