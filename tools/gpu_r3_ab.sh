#!/bin/bash
# short A/B runs of the contract bench: env var settings given as arguments, one run each
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --steps 6 --warmup 2 --no-legs --no-cpu-baseline --no-pmc > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$cfg" <<'PY'
import json, sys
d=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{')][-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()}, d['check_groups']['status'])
PY
done
