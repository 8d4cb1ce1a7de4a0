#!/bin/bash
# libmem serves large requests from larger parked bins: its own tests, the verify-clean sweeps, then the cold leg's allocations
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
tmp=$(mktemp -d)
ARGS="--leg --cold --rows 1e9 --batch-rows 67108864 --steps 3 --warmup 1"
ARES_RTC_CACHE_DIR=$tmp timeout 60 python bench.py $ARGS > gpurun_out/r4/memfit_fill.json 2>/dev/null; echo "fill rc $?"
rm -f gpurun_out/r4/memfit.trace
ARES_RTC_TRACE=$PWD/gpurun_out/r4/memfit.trace ARES_RTC_CACHE_DIR=$tmp timeout 60 python bench.py $ARGS > gpurun_out/r4/memfit.json 2>/dev/null; echo "warm rc $?"
python -c "
import json
for f in ('memfit_fill', 'memfit'):
    d = json.load(open('gpurun_out/r4/%s.json' % f)); print(f, round(d['cold_first_query_ms'], 1), [round(x, 1) for x in d['cold_first_query_batch_ms']][:3], 'new constants', round(d['new_constants_query_ms'], 1), [round(x, 1) for x in d['new_constants_batch_ms']][:3], d['cold_check_groups'], d['new_constants_check_groups'])"
grep "driver\|after a wait" gpurun_out/r4/memfit.trace | cut -c1-170
timeout 95 python -m pytest tests/test_expand_libmem.py tests/test_stream_lifetime.py tests/test_write_tracking.py tests/test_executor.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
