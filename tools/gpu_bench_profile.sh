#!/bin/bash
# Runs on the GPU box (via gpurun): the contract bench (all legs), a rocprofv3 kernel-trace + stats of
# the same command without the secondary legs, and the PMC passes.  Outputs land in gpurun_out/.
#   bash tools/gpu_bench_profile.sh TAG
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r2}
mkdir -p $OUT
cd $R
timeout 1200 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench exit $?"
tail -c 1500 $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
export TMPDIR=/tmp
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python $R/bench.py --no-cpu-baseline --no-legs > $OUT/prof_$TAG.log 2>&1; echo "rocprof exit $?"
f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "ares::|_rtc" $f | cut -c1-200 | head -14
find $OUT/prof_$TAG -name '*kernel_trace.csv' -delete
cd $R
bash tools/gpu_pmc.sh $TAG 67108864 > $OUT/pmc_$TAG.log 2>&1; echo "pmc exit $?"
