#!/bin/bash
# LDS / wait counters of the bench kernels (scratch output under gpurun_out/pmc_lds*)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-noskip --steps 5 --warmup 2"
rm -rf $O/pmc_lds1 $O/pmc_lds2
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_lds1 -- $B > $O/pmc_lds1.log 2>&1 && echo lds1_ok &&
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_lds2 -- $B > $O/pmc_lds2.log 2>&1 && echo lds2_ok
find $O/pmc_lds1 $O/pmc_lds2 -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
tail -3 $O/pmc_lds1.log $O/pmc_lds2.log
