#!/bin/bash
# which of a cold process's large DeviceAllocate calls go to the driver, and why
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
tmp=$(mktemp -d)
ARGS="--leg --cold --rows 1e9 --batch-rows 67108864 --steps 3 --warmup 1"
ARES_RTC_CACHE_DIR=$tmp timeout 200 python bench.py $ARGS > /dev/null 2>&1
rm -f gpurun_out/r4/memtrace.trace
ARES_RTC_TRACE=$PWD/gpurun_out/r4/memtrace.trace ARES_RTC_CACHE_DIR=$tmp timeout 200 python bench.py $ARGS > gpurun_out/r4/memtrace.json 2>/dev/null
python -c "
import json; d = json.load(open('gpurun_out/r4/memtrace.json')); print(round(d['cold_first_query_ms'], 1), [round(x, 1) for x in d['cold_first_query_batch_ms']][:4], round(d['new_constants_query_ms'], 1), [round(x, 1) for x in d['new_constants_batch_ms']][:4])"
grep -v "CreateCudaStream\|disk_read\|scratch_bytes\|module_load" gpurun_out/r4/memtrace.trace | cut -c1-200
