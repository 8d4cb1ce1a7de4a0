#!/bin/bash
# whole GPU suite, then a short contract bench (no legs), then optionally more
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
tag=${1:-a}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4/suite_$tag.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4/suite_$tag.log
tail -5 gpurun_out/r4/suite_$tag.log
grep -E "^FAILED|^ERROR" gpurun_out/r4/suite_$tag.log | head -20
timeout 400 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-pmc > gpurun_out/r4/bench_$tag.json 2> gpurun_out/r4/bench_$tag.err
echo "bench rc $?"
python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/r4/bench_{sys.argv[1]}.json') if l.startswith('{')][-1])
    print('ms/step', round(d['ms_per_step'], 3), 'median', round(d['median_ms_per_step'], 3), 'value', round(d['value'] / 1e9, 2), 'G rows/s', d['check_groups']['status'])
    print({k: round(v['avg_ms'], 4) for k, v in d['kernels'].items()})
    print(d['roofline'])
except Exception as e:
    print('no bench line', e)
PY
tail -3 gpurun_out/r4/bench_$tag.err
