#!/bin/bash
# the whole GPU suite on the final tree, then the headline + the cold-start legs
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4/suite_last.log 2>&1
echo "pytest rc $? $(tail -1 gpurun_out/r4/suite_last.log)"
grep -E "^FAILED|^ERROR" gpurun_out/r4/suite_last.log | head
ARES_RTC_TRACE=$PWD/gpurun_out/r4/last.trace timeout 300 python bench.py --steps 10 --warmup 3 --legs cold,live --no-cpu-baseline --no-pmc > gpurun_out/r4/bench_last.json 2> gpurun_out/r4/bench_last.err
echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4/bench_last.json') if l.startswith('{')][-1])
print('ms/step', round(d['ms_per_step'], 3), 'median', round(d['median_ms_per_step'], 3), d['check_groups']['status'], {k: round(v['avg_ms'], 4) for k, v in d['kernels'].items()})
for k, v in d['legs'].items():
    if isinstance(v, dict):
        print(k, {a: (b if not isinstance(b, list) else [round(x, 1) for x in b][:4]) for a, b in v.items() if a in ('cold_first_query_ms', 'cold_first_query_batch_ms', 'new_constants_query_ms', 'new_constants_batch_ms', 'warm_query_ms', 'ms_per_step', 'check_groups', 'cold_check_groups')})
PY
grep -c . gpurun_out/r4/last.trace; grep "hipMalloc of" gpurun_out/r4/last.trace | cut -c1-120
