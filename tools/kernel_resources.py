#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy of every kernel of the C-ABI library, as the compiler reports them
(`-Rpass-analysis=kernel-resource-usage`, the flags of csrc/Makefile, gfx950; no GPU needed).

    python tools/kernel_resources.py                 # table on stdout
    python tools/kernel_resources.py --json out.json # {kernel: {...}}

Used by tests/test_kernel_resources.py (the limits of the hot kernels are pinned there, so an occupancy cliff or a spill
shows at build time) and by tools/kernel_table.py (DESIGN.md's generated table).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vk-renderer_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DVKR_CONTRACT=2",
         "-fno-gpu-flush-denormals-to-zero", "-Wno-unused-function", "--cuda-device-only", "-S",
         "-Rpass-analysis=kernel-resource-usage"]

FIELDS = {
    "TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes",
    "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
    "LDS Size [bytes/block]": "lds_bytes",
}


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool or not names:
        return names
    out = subprocess.run([tool] + names, capture_output=True, text=True)
    if out.returncode != 0:
        return names
    return out.stdout.strip().split("\n")


def short(d):
    """'void vkr::k_gtao_main<true, false>(vkr::GtaoArgs)' -> 'k_gtao_main<true, false>'"""
    d = re.sub(r"^void ", "", d)
    d = d.replace("vkr::", "")
    depth = 0
    for i, c in enumerate(d):
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            return d[:i]
    return d


def resources(sources=None):
    """{kernel (short, demangled): {file, sgprs, vgprs, scratch_bytes, occupancy, sgpr_spill, vgpr_spill, lds_bytes}}"""
    if sources is None:
        sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for f in sources:
            cmd = [HIPCC] + FLAGS + ["-o", os.path.join(tmp, f + ".s"), os.path.join(CSRC, f)]
            procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for f, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError(f"{f}: hipcc failed\n{out[-3000:]}")
            cur = None
            for line in out.split("\n"):
                m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
                if not m:
                    continue
                body = m.group(1).strip()
                if body.startswith("Function Name:"):
                    cur = body.split(":", 1)[1].strip()
                    res[cur] = {"file": f}
                elif cur and ":" in body:
                    k, v = body.rsplit(":", 1)
                    if k.strip() in FIELDS:
                        res[cur][FIELDS[k.strip()]] = int(v)
    names = list(res)
    dem = demangle(names)
    return {short(d): res[n] for n, d in zip(names, dem)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("sources", nargs="*")
    a = ap.parse_args()
    r = resources(a.sources or None)
    if a.json:
        with open(a.json, "w") as g:
            json.dump(r, g, indent=1, sort_keys=True)
    print(f"{'kernel':58s} {'VGPR':>4s} {'SGPR':>4s} {'scratch':>7s} {'LDS':>6s} {'occ':>3s}  spills(v/s)")
    for k in sorted(r):
        v = r[k]
        print(f"{k:58s} {v.get('vgprs', -1):4d} {v.get('sgprs', -1):4d} {v.get('scratch_bytes', -1):7d} {v.get('lds_bytes', -1):6d} "
              f"{v.get('occupancy', -1):3d}  {v.get('vgpr_spill', 0)}/{v.get('sgpr_spill', 0)}")


if __name__ == "__main__":
    sys.exit(main())
