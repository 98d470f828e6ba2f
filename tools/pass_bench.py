#!/usr/bin/env python3
"""Time single passes of the chain back-to-back (no other pass in between, one event pair around N launches):
the iteration loop for kernel work.  Needs a GPU.

    python tools/pass_bench.py [--frame 3840x2160] [--iters 200] [--passes taa,gtao_filter,...]

The frame is brought to steady state first (two whole frames), so every pass sees the inputs it sees in bench.py.
Prints one line per pass: mean launch time in ms.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from vk_renderer_amd.chain import PostFxChain  # noqa: E402

PASSES = ("downsample", "ssr_trace", "ssr_filter", "ssr_blur", "gtao_main", "gtao_filter", "gtao_accumulate", "taa")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frame", default="3840x2160")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--passes", default=",".join(PASSES))
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    w, h = (int(v) for v in args.frame.split("x"))
    chain = PostFxChain(w, h, backend="product", device="cuda")
    chain.synth()
    chain.build_prev_hiz()
    chain.init_histories()
    chain.preintegrate_pdf()
    for _ in range(2):
        chain.frame()
        chain.swap_histories()
    chain.frame()
    chain.sync()
    out = {}
    for name in args.passes.split(","):
        fn = getattr(chain, name)
        for _ in range(5):
            fn()
        chain.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        e1.synchronize()
        out[name] = e0.elapsed_time(e1) / args.iters
        print(f"{name:16s} {out[name]:.4f} ms")
    print(f"sum              {sum(out.values()):.4f} ms")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main()
