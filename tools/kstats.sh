#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, prints the per-kernel mean durations.
#   bash tools/kstats.sh [name] [extra bench.py flags...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-kstats}; shift || true
O=$R/gpurun_out/$N
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-noskip --steps 20 --warmup 3 "$@" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
find $O -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
python3 - "$O" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Name'].split('(')[0]
        if 'vkr::' in n:
            print(f"{n[:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:.4f} ms")
PY
