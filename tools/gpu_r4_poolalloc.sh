#!/bin/bash
# fresh blocks from the stream-ordered pool (ARES_TEMP_POOL_ALLOC / ARES_MEM_POOL_ALLOC): the empty-cache cold leg with and without
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
ARGS="--leg --cold --rows 1e9 --batch-rows 67108864 --steps 3 --warmup 1"
show() { python -c "
import json; d = json.load(open('gpurun_out/r4/$1.json')); print('$1', round(d['cold_first_query_ms'], 1), [round(x, 1) for x in d['cold_first_query_batch_ms']][:3], 'warm', round(d['warm_query_ms'], 2), 'new constants', round(d['new_constants_query_ms'], 1), d['cold_check_groups'], d['new_constants_check_groups'])"; }
for tag in base pool base2 pool2; do
  tmp=$(mktemp -d)
  if [ ${tag:0:4} = pool ]; then export ARES_TEMP_POOL_ALLOC=1 ARES_MEM_POOL_ALLOC=1; else unset ARES_TEMP_POOL_ALLOC ARES_MEM_POOL_ALLOC; fi
  ARES_RTC_CACHE_DIR=$tmp timeout 25 python bench.py $ARGS > gpurun_out/r4/pa_$tag.json 2>gpurun_out/r4/pa_$tag.err; echo "$tag rc $?"; show pa_$tag
done
