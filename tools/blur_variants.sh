#!/bin/bash
# GPU box: time the blur's launch orders (VKR_BLUR_QUEUE = 0 plain, 1 heaviest first per XCD, 2 heavy / light interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
for v in 0 1 2; do
  VKR_BLUR_QUEUE=$v python3 $R/bench.py --steps 50 --no-cpu-baseline > $O/blur_q$v.json 2> $O/blur_q$v.err || exit 1
  python3 - <<PY
import json
d=json.load(open("$O/blur_q$v.json"))
print("queue $v: step %.4f ms  blur %.4f ms  (live %.4f)" % (d["ms_per_step"], d["per_pass_ms"]["SSSR_blur"], d["roofline"]["avg_launch_ms"] if d["roofline"]["kernel"]=="SSSR_blur" else -1))
PY
done
