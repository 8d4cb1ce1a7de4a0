#!/bin/bash
# the driver's bench command alone, summary printed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3close.json 2> gpurun_out/bench_r3close.err; echo "bench exit $? after $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r3close.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step','max_ms_per_step','wall_s_whole_run')}, d['check_groups']['status'])
print('roofline', {k:v for k,v in d['roofline'].items() if k!='note'})
for k,v in d['legs'].items():
    if isinstance(v, list): print(k, [(r.get('ms'), r.get('leg_wall_s')) for r in v]); continue
    if not isinstance(v, dict): print(k, v); continue
    print(k, {a:v[a] for a in ('ms_per_step','leg_wall_s','skipped','error','cold_first_query_ms','check_groups') if a in v})
PY
