#!/usr/bin/env python3
"""Puts the GENERATED tables into DESIGN.md between their markers (nothing numeric in those blocks is typed by hand):

  <!-- BEGIN GENERATED: kernel_table -->  ... profiles/<tag>_kernel_table.md (tools/kernel_table.py)          <!-- END GENERATED: kernel_table -->
  <!-- BEGIN GENERATED: bench_table -->   ... one row per committed bench line profiles/<tag>_bench*.json      <!-- END GENERATED: bench_table -->

    python tools/design_tables.py --tag r04_final
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ROWS = [  # (file suffix, label)
    ("", "**c2** 3840x2160 composite (the metric), frozen scene"),
    ("_textured", "c2 with roughness per texel (`--material textured`)"),
    ("_c1", "c1 1920x1080 GTAO main only, non-MIS"),
    ("_c3", "c3 7680x4320 composite"),
    ("_c4_n1", "c4 15360x8640 composite on ONE GPU (the denominator of the scaling curve)"),
    ("_c5", "c5 3840x2160, 8 x (trace, filter, blur) + TAA"),
    ("_tiled1", "c2 through the tiled code path on one rank (`--rehearse-tiled`)"),
    ("_shading", "c2 + deferred shading (`--shading`)"),
    ("_raster", "c2 with the G-buffer rasterised every frame (`--raster`, 37 k triangles)"),
]


def bench_table(tag):
    lines = ["| config (`profiles/" + tag + "_bench*.json`) | ms per frame | median of the event-timed frames | Gpx/s | skips off (`value_noskip`) | skips off and per-lane blur (`value_generic`) |",
             "|---|---|---|---|---|---|"]
    for suffix, label in ROWS:
        path = os.path.join(ROOT, "profiles", f"{tag}_bench{suffix}.json")
        if not os.path.exists(path) and suffix == "_c4_n1":
            path = os.path.join(ROOT, "profiles", "r04_bench_c4_n1.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            d = json.load(f)
        med = f"{d['ms_per_step_median']:.3f}" if "ms_per_step_median" in d else "-"
        ns = f"{d['ms_per_step_noskip']:.3f} ms / {d['value_noskip'] / 1e3:.2f} Gpx/s" if "value_noskip" in d else "-"
        ge = f"{d['ms_per_step_generic']:.3f} ms / {d['value_generic'] / 1e3:.2f} Gpx/s" if "value_generic" in d else "-"
        lines.append(f"| {label} | {d['ms_per_step']:.3f} | {med} | {d['value'] / 1e3:.2f} | {ns} | {ge} |")
    path = os.path.join(ROOT, "profiles", f"{tag}_bench.json")
    if os.path.exists(path):
        with open(path) as f:
            d = json.load(f)
        r, c = d["roofline"], d.get("cpu_baseline")
        lines += ["", f"The c2 line's roofline object: `{r['kernel']}` {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB / {r['avg_launch_ms']:.4f} ms = {r['achieved']:.0f} GB/s = "
                  f"**{r['frac']:.4f} of the 8 TB/s peak**, {r['frac_of_measured']:.3f} of the {d['measured_read_gbps'] / 1e3:.2f} TB/s float4 stream read measured in the same run; "
                  f"composite {d['composite_gbps']:.0f} GB/s."]
        if c:
            lines.append(f"CPU restatement in the same run (`cpu_baseline`, kind \"{c['kind']}\"): {c['value']:.2f} Mpx/s on {c['cores']} cores ({c['sample']}).")
    return "\n".join(lines)


def inject(text, name, body):
    a, b = f"<!-- BEGIN GENERATED: {name} -->", f"<!-- END GENERATED: {name} -->"
    if a not in text or b not in text:
        raise SystemExit(f"DESIGN.md has no {a} ... {b} block")
    return re.sub(re.escape(a) + r".*?" + re.escape(b), lambda m: a + "\n" + body + "\n" + b, text, flags=re.S)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    a = ap.parse_args()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "kernel_table.py"), "--tag", a.tag], stdout=subprocess.DEVNULL)
    with open(os.path.join(ROOT, "profiles", f"{a.tag}_kernel_table.md")) as f:
        kt = f.read().strip()
    kt = "\n".join(l for l in kt.split("\n") if not l.startswith("# "))
    p = os.path.join(ROOT, "DESIGN.md")
    with open(p) as f:
        text = f.read()
    text = inject(text, "kernel_table", kt.strip())
    text = inject(text, "bench_table", bench_table(a.tag))
    with open(p, "w") as f:
        f.write(text)
    print("DESIGN.md: kernel_table and bench_table regenerated from profiles/" + a.tag + "_*")


if __name__ == "__main__":
    main()
