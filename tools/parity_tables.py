#!/usr/bin/env python3
"""Assemble the per-test parity tables the GPU tests write (`parity_table` fixture -> gpurun_out/parity_test_*.json)
into the three per-configuration files committed under profiles/.

  python tools/parity_tables.py --round r03
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOTE = ("rows: every comparison the test made (tests/parity.py): bit-exact for the integer surfaces; otherwise a texel is outside tolerance "
        "when |d| > 1e-3 |ref| AND |d| > one storage step of its format.  MI355X, {round}, numeric contract 2.")
FILES = {
    "parity_c2.json": ("c2 3840x2160", [
        ("stagewise (each pass fed the oracle's bytes)", "parity_test_chain_stagewise_full_size.json"),
        ("end_to_end (two frames, no resynchronisation)", "parity_test_chain_end_to_end_size1_.json")]),
    "parity_c3.json": ("c3 7680x4320", [
        ("synthetic G-buffer, stagewise", "parity_test_c3_8k_synthetic_stagewise.json"),
        ("rasterised procedural scene through SceneRenderer, then one frame", "parity_test_c3_8k_rasterised_through_scene_renderer.json")]),
    "parity_c5.json": ("c5 3840x2160, 8 x (trace, filter, blur) + TAA through the host mirror", [
        ("rows", "parity_test_c5_eight_rays_per_pixel_loop.json")]),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    a = ap.parse_args()
    for out, (config, parts) in FILES.items():
        doc = {"config": config, "note": NOTE.format(round=a.round)}
        for title, src in parts:
            with open(os.path.join(ROOT, "gpurun_out", src)) as f:
                t = json.load(f)
            doc[title] = t["rows"]
            doc.setdefault("tests", []).append(t["test"])
        with open(os.path.join(ROOT, "profiles", out), "w") as f:
            json.dump(doc, f, indent=1)
        total = sum(r["outside_tolerance"] for title, _ in parts for r in doc[title])
        print(out, "texels outside tolerance:", total)
