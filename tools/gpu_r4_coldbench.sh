#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
timeout 600 python bench.py --steps 3 --warmup 1 --legs cold --no-cpu-baseline --no-pmc > gpurun_out/r4/coldbench.json 2> gpurun_out/r4/coldbench.err
echo rc $?
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4/coldbench.json') if l.startswith('{')][-1])
for k in ('cold_process', 'cold_process_warm_disk_cache'):
    v = d['legs'].get(k)
    print(k, {a: (b if not isinstance(b, list) else b[:4] + b[-1:]) for a, b in v.items() if 'ms' in a} if isinstance(v, dict) else v)
print('ms/step', d['ms_per_step'])
PY
