#!/bin/bash
# On the GPU box: a few PMC passes of bench.py (memory-pipe and lane-utilisation counters), per-kernel means printed.
#   bash tools/pmc_probe.sh "CTR1 CTR2 ..." ["CTR3 ..."] ...      (one rocprofv3 --pmc pass per argument)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1)); O=$R/gpurun_out/pmcp_$i; rm -rf $O
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-noskip --steps 5 --warmup 2 > $O.log 2>&1 || { echo "pass $i failed"; tail -5 $O.log; continue; }
  python3 - "$O" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'vkr::k_' in n and ('sssr' in n or 'gtao_main' in n or 'taa' in n):
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in sorted(acc.items()):
    print(n[:28].ljust(28), '  '.join(f"{c} {sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
  find $O -type f -delete 2>/dev/null
done
