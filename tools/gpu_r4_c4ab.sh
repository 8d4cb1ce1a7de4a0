#!/bin/bash
# A/B of the sort path on the small C4 configuration (64 Mi rows, 200 k groups): env settings as arguments
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
python -c "import torch" 2>/dev/null
for cfg in "$@"; do
  env $(echo $cfg | tr ',' ' ') timeout 300 python tools/bench_configs.py c4 > gpurun_out/r4/c4ab.json 2> gpurun_out/r4/c4ab.err
  python - "$cfg" <<'PY'
import json, sys
try:
    for l in open('gpurun_out/r4/c4ab.json'):
        if l.startswith('{'):
            d = json.loads(l)
            print(sys.argv[1], 'ms', round(d['ms'], 3), 'sum_ok', d.get('sum_ok'), {k: v for k, v in d['kernels'].items() if k.startswith(('radix', 'sort', 'reduce', 'hash_lookup'))})
except Exception as e:
    print(sys.argv[1], 'failed', e, open('gpurun_out/r4/c4ab.err').read()[-300:])
PY
done
