#!/bin/bash
# whole GPU suite, then the contract bench with selected legs
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
tag=${1:-b}
legs=${2:-time_filters}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r4/suite_$tag.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4/suite_$tag.log
tail -4 gpurun_out/r4/suite_$tag.log
grep -E "^FAILED|^ERROR|^E  " gpurun_out/r4/suite_$tag.log | head -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --legs $legs --leg-budget 0 > gpurun_out/r4/bench_$tag.json 2> gpurun_out/r4/bench_$tag.err
echo "bench rc $?"
python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/r4/bench_{sys.argv[1]}.json') if l.startswith('{')][-1])
    print('ms/step', round(d['ms_per_step'], 3), 'median', round(d['median_ms_per_step'], 3), 'value', round(d['value'] / 1e9, 2), 'G rows/s', d['check_groups']['status'])
    print({k: round(v['avg_ms'], 4) for k, v in d['kernels'].items()})
    print('frac', round(d['roofline']['frac'], 3), 'all kernels', round(d['roofline_all_kernels']['frac'], 3))
    for name, leg in d['legs'].items():
        if isinstance(leg, dict) and 'ms_per_step' in leg:
            print(name, 'ms/step', round(leg['ms_per_step'], 3), leg['check_groups'], {k: round(v['avg_ms'], 4) for k, v in leg['kernels'].items()})
        else:
            print(name, str(leg)[:400])
except Exception as e:
    print('no bench line', e)
PY
tail -3 gpurun_out/r4/bench_$tag.err
if [ -n "$3" ]; then
  timeout 700 python tools/bench_configs.py $3 > gpurun_out/r4/configs_$tag.json 2> gpurun_out/r4/configs_$tag.err
  echo "bench_configs $3 rc $?"
  cut -c1-1500 gpurun_out/r4/configs_$tag.json
  tail -2 gpurun_out/r4/configs_$tag.err
fi
