#!/bin/bash
# short A/B runs of the contract bench: env var settings given as arguments (comma separated inside one argument), one run each
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
python -c "import torch" 2>/dev/null
for cfg in "$@"; do
  env $(echo $cfg | tr ',' ' ') timeout 300 python bench.py --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-pmc ${BENCH_ARGS} > gpurun_out/r4/ab.json 2> gpurun_out/r4/ab.err
  python - "$cfg" <<'PY'
import json, sys
try:
    d=json.loads([l for l in open('gpurun_out/r4/ab.json') if l.startswith('{')][-1])
    print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()}, d['check_groups']['status'])
except Exception as e:
    print(sys.argv[1], 'failed', e, open('gpurun_out/r4/ab.err').read()[-300:])
PY
done
