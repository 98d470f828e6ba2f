#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of bench.py.
# Output lands in gpurun_out/<name>; digest with tools/profile_digest.py afterwards.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-noskip"
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sqa $O/pmc_sqb
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B --steps 20 --warmup 3 > $O/prof.log 2>&1 && echo stats_ok &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1 && echo fetch_ok &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B --steps 5 --warmup 2 > $O/pmc_write.log 2>&1 && echo write_ok &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sqa -- $B --steps 5 --warmup 2 > $O/pmc_sqa.log 2>&1 && echo sqa_ok &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_sqb -- $B --steps 5 --warmup 2 > $O/pmc_sqb.log 2>&1 && echo sqb_ok
# keep only the csv files that are digested (the merge-back limit is 64 MiB)
find $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sqa $O/pmc_sqb -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
du -sh $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sqa $O/pmc_sqb
