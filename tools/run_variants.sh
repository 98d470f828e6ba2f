#!/bin/bash
# On the GPU box: bench.py once per variant built by tools/build_variants.sh; prints step and per-pass times.
#   bash tools/run_variants.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do
  V=$R/vk-renderer_amd/csrc/variants/$n
  LD_LIBRARY_PATH=$V:${LD_LIBRARY_PATH:-} VKR_POSTFX_LIB=$V/libvkr_postfx.so python3 $R/bench.py --no-cpu-baseline --no-noskip --steps 100 > $R/gpurun_out/var_$n.json 2> $R/gpurun_out/var_$n.err || { echo "$n FAILED"; tail -3 $R/gpurun_out/var_$n.err; continue; }
  python3 - "$n" "$R/gpurun_out/var_$n.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
pp = d["per_pass_ms"]
print(f"{sys.argv[1]:12s} step {d['ms_per_step']:.4f} median {d.get('ms_per_step_median', 0):.4f}  " + "  ".join(f"{k.replace('SSSR_', '').replace('GTAO_', 'g_')[:9]} {v:.4f}" for k, v in pp.items()))
PY
done
