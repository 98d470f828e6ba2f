#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of bench.py for one BASELINE config.
#   bash tools/profile_run.sh [config = c2] [extra bench.py arguments]
# Output lands in gpurun_out/{prof,pmc_fetch,pmc_write,pmc_sqa,pmc_sqb}_<config>; digest with
#   python tools/profile_digest.py --tag <tag> --config <config> --frame WxH --stats gpurun_out/prof_<config> ...
# The program after `--` is python itself (never env / bash -c: the profiler's preloaded library has initialised the GPU).
set -u
CFG=${1:-c2}
shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config $CFG --no-cpu-baseline --no-noskip $*"
S=20; P=5
if [ "$CFG" = "c4" ]; then S=8; P=3; fi
D="$O/prof_$CFG $O/pmc_fetch_$CFG $O/pmc_write_$CFG $O/pmc_sqa_$CFG $O/pmc_sqb_$CFG"
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$CFG -- $B --steps $S --warmup 3 > $O/prof_$CFG.log 2>&1 && echo stats_ok &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$CFG -- $B --steps $P --warmup 2 > $O/pmc_fetch_$CFG.log 2>&1 && echo fetch_ok &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$CFG -- $B --steps $P --warmup 2 > $O/pmc_write_$CFG.log 2>&1 && echo write_ok &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sqa_$CFG -- $B --steps $P --warmup 2 > $O/pmc_sqa_$CFG.log 2>&1 && echo sqa_ok &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_sqb_$CFG -- $B --steps $P --warmup 2 > $O/pmc_sqb_$CFG.log 2>&1 && echo sqb_ok &&
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sqc_$CFG -- $B --steps $P --warmup 2 > $O/pmc_sqc_$CFG.log 2>&1 && echo sqc_ok
# keep only the csv files that are digested (the merge-back limit is 64 MiB)
find $D $O/pmc_sqc_$CFG -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
du -sh $D $O/pmc_sqc_$CFG
