#!/usr/bin/env python3
"""Digest rocprofv3 output directories (scratch, under gpurun_out/) into the small summaries
committed under profiles/.

  python tools/profile_digest.py --tag r01_final \
      --stats gpurun_out/prof --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write \
      --sq gpurun_out/pmc_sq gpurun_out/pmc_sq2

Writes profiles/<tag>_kernel_stats.csv (the --stats table restricted to vkr:: kernels),
profiles/<tag>_pmc_traffic.csv + profiles/traffic.json (HBM bytes per task, corrected as
MI355X_MICROARCH.md prescribes: read bytes = 2 x FETCH_SIZE on gfx950, WRITE_SIZE exact, both
in KB), and profiles/<tag>_sq_counters.md (per-kernel means of every SQ counter collected).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

TASK_OF = {  # kernel -> (rendergraph task, launches of that kernel per task)
    "vkr::k_downsample_gbuffer": ("DownsampleGbuffer", 1),
    "vkr::k_depth_mips_fused": ("DownsampleDepth", 2),
    "vkr::k_sssr_trace": ("SSSR_trace", 1),
    "vkr::k_sssr_trace_resume": ("SSSR_trace", 1),  # the second launch of the same task (vkr_sssr_trace_split)
    "vkr::k_sssr_filter": ("SSSR_filter", 1),
    "vkr::k_sssr_blur": ("SSSR_blur", 1),
    "vkr::k_gtao_main": ("GTAO_main", 1),
    "vkr::k_gtao_filter": ("GTAO_filter", 1),
    "vkr::k_gtao_accumulate": ("GTAO_accumulate", 1),
    "vkr::k_taa_resolve": ("TAA", 1),
    "vkr::k_defered_shading": ("DeferedShading", 1),
}


def short(name):
    name = name.strip('"')
    for pre in ("void ",):
        if name.startswith(pre):
            name = name[len(pre):]
    return name.split("(")[0].split("<")[0]


def find(d, suffix):
    # gpurun merges every call's files into the same local directory: take the newest, not the last by name
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[-1]


def counters(d, skip_first=2):
    """{kernel: {counter: mean over dispatches}} — the first dispatches of each kernel (warm-up) are dropped."""
    per = defaultdict(lambda: defaultdict(list))
    with open(find(d, "counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if not k.startswith("vkr::"):
                continue
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "SQ_ACTIVE_INST_VALU":
                per[k]["_duration_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    out = {}
    for k, cs in per.items():
        out[k] = {}
        for c, vals in cs.items():
            per_task = TASK_OF.get(k, (k, 1))[1]
            vals = vals[skip_first * per_task:] or vals
            out[k][c] = (sum(vals) / len(vals), len(vals))
    return out


def library_sha16():
    """which build of the kernels the profiled command ran: sha256 of csrc/libvkr_postfx.so (the file travels to the GPU box as it
    is), so that bench.py can say whether a committed digest belongs to the library it has loaded"""
    import hashlib

    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vk-renderer_amd", "csrc", "libvkr_postfx.so")
    if not os.path.exists(lib):
        return None
    with open(lib, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def merge_by_config(path, config, digest):
    """profiles/traffic.json / valu_busy.json: {config: {task: value, "_source": ...}} — one digest per BASELINE config"""
    table = {}
    if os.path.exists(path):
        with open(path) as f:
            table = json.load(f)
        if "_source" in table:  # the round-3 layout: one flat digest, of c2
            table = {"c2": table}
    digest["_library_sha16"] = library_sha16()
    table[config] = digest
    with open(path, "w") as g:
        json.dump(table, g, indent=1, sort_keys=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq", nargs="*", default=[])
    ap.add_argument("--note", default="")
    ap.add_argument("--config", default="c2", help="BASELINE config the profiled command ran (bench.py --config): the key under which "
                    "profiles/traffic.json and profiles/valu_busy.json keep this digest")
    ap.add_argument("--frame", default="3840x2160")
    a = ap.parse_args()
    os.makedirs("profiles", exist_ok=True)

    if a.stats:
        src = find(a.stats, "kernel_stats.csv")
        with open(src) as f, open(f"profiles/{a.tag}_kernel_stats.csv", "w") as g:
            g.write(f"# rocprofv3 --kernel-trace --stats; {a.note}\n")
            for i, line in enumerate(f):
                if i == 0 or "vkr::" in line:
                    g.write(line)
        print("wrote", f"profiles/{a.tag}_kernel_stats.csv")

    if a.fetch and a.write:
        fe, wr = counters(a.fetch), counters(a.write)
        traffic = {}
        with open(f"profiles/{a.tag}_pmc_traffic.csv", "w") as g:
            g.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); {a.note}\n")
            cal = fe.get("vkr::k_stream_read", {}).get("FETCH_SIZE")
            if cal:
                g.write(f"# calibration: k_stream_read FETCH_SIZE {cal[0]:.1f} KB per launch; read bytes = 2 x FETCH_SIZE "
                        "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact\n")
            g.write("task,kernel,dispatches_sampled,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,launches_per_task,corrected_bytes_per_task\n")
            for k, (task, n) in TASK_OF.items():
                if k not in fe or "FETCH_SIZE" not in fe[k] or k not in wr:
                    continue
                f_kb, cnt = fe[k]["FETCH_SIZE"]
                w_kb, _ = wr[k]["WRITE_SIZE"]
                b = (2.0 * f_kb + w_kb) * 1024.0 * n
                traffic[task] = traffic.get(task, 0.0) + b  # a task of several kernels: their sum
                g.write(f"{task},{k},{cnt},{f_kb:.1f},{w_kb:.1f},{n},{b:.0f}\n")
        traffic["_source"] = f"{a.tag}_pmc_traffic.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `bench.py --config {a.config}` at {a.frame}, bytes = 2 x FETCH_SIZE KB + WRITE_SIZE KB per task"
        merge_by_config("profiles/traffic.json", a.config, traffic)
        print("wrote", f"profiles/{a.tag}_pmc_traffic.csv", "profiles/traffic.json")

    if a.sq:
        merged = defaultdict(dict)
        for d in a.sq:
            for k, cs in counters(d).items():
                if "SQ_THREAD_CYCLES_VALU" in cs and "SQ_ACTIVE_INST_VALU" in cs:  # both from ONE pass: the ratio is the lane use
                    merged[k]["_lanes"] = cs["SQ_THREAD_CYCLES_VALU"][0] / cs["SQ_ACTIVE_INST_VALU"][0]
                for c, (v, _) in cs.items():
                    merged[k][c] = v
        names = sorted({c for cs in merged.values() for c in cs if c != "_lanes"})
        with open(f"profiles/{a.tag}_sq_counters.md", "w") as g:
            g.write(f"# SQ counters per kernel launch (means), rocprofv3 --pmc passes; {a.note}\n\n")
            g.write("| kernel | " + " | ".join(names) + " | VALU busy (4·ACTIVE_INST_VALU / 1024 SIMD / (duration·2.4 GHz)) | active lanes per VALU instruction (THREAD_CYCLES_VALU / ACTIVE_INST_VALU, same pass) |\n|---|" + "---|" * (len(names) + 2) + "\n")
            for k in TASK_OF:
                if k not in merged:
                    continue
                cs = merged[k]
                ratio = ""
                if "SQ_ACTIVE_INST_VALU" in cs and cs.get("_duration_ns"):
                    # gfx94x VALUBusy formula: 4 * SQ_ACTIVE_INST_VALU / SIMD_NUM / cycles, 1024 SIMDs, 2.4 GHz
                    ratio = f"{4.0 * cs['SQ_ACTIVE_INST_VALU'] / 1024.0 / (cs['_duration_ns'] * 2.4):.2f}"
                lanes = f"{cs['_lanes']:.1f}" if "_lanes" in cs else ""
                g.write(f"| {k} | " + " | ".join(f"{cs.get(c, float('nan')):.4g}" for c in names) + f" | {ratio} | {lanes} |\n")
        print("wrote", f"profiles/{a.tag}_sq_counters.md")
        busy, weight = {}, {}
        for k, (task, _) in TASK_OF.items():
            cs = merged.get(k, {})
            if "SQ_ACTIVE_INST_VALU" in cs and cs.get("_duration_ns"):
                # a task of several kernels: duration-weighted
                busy[task] = busy.get(task, 0.0) + 4.0 * cs["SQ_ACTIVE_INST_VALU"] / 1024.0 / 2.4
                weight[task] = weight.get(task, 0.0) + cs["_duration_ns"]
        busy = {t: v / weight[t] for t, v in busy.items()}
        busy["_source"] = f"{a.tag}_sq_counters.md: rocprofv3 --pmc SQ_ACTIVE_INST_VALU of `bench.py --config {a.config}` at {a.frame}, 4 x count / 1024 SIMDs / (duration x 2.4 GHz)"
        merge_by_config("profiles/valu_busy.json", a.config, busy)
        print("wrote profiles/valu_busy.json")


if __name__ == "__main__":
    main()
