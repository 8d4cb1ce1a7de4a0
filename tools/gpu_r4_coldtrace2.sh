#!/bin/bash
# second pass of tools/gpu_r4_coldtrace.sh: every checked runtime call is timed now
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4
tmp=$(mktemp -d)
ARGS="--leg --cold --rows 1e9 --batch-rows 67108864 --steps 3 --warmup 1"
for run in fill warm1 warm2 warm3; do
  rm -f $R/gpurun_out/r4/ct2_$run.trace
  ARES_RTC_TRACE=$R/gpurun_out/r4/ct2_$run.trace ARES_RTC_CACHE_DIR=$tmp $EXTRA timeout 300 python bench.py $ARGS > gpurun_out/r4/ct2_$run.json 2>gpurun_out/r4/ct2_$run.err
  echo "== $run rc $?"; python -c "
import json; d = json.load(open('gpurun_out/r4/ct2_$run.json')); print('  first query', round(d['cold_first_query_ms'], 1), [round(x, 1) for x in d['cold_first_query_batch_ms']][:6])"
  grep -v "CreateCudaStream\|disk_read\|scratch_bytes" gpurun_out/r4/ct2_$run.trace | cut -c1-160
done
