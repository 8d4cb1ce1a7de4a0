#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_scale_parity.py > gpurun_out/r2b_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r2b_pytest.log
tail -15 gpurun_out/r2b_pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-legs --no-cpu-baseline > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2b_bench.json'))
print(d['value'], d['ms_per_step'], d['check_groups'])
for k,v in d['kernels'].items(): print(k, v)
PY
tail -5 gpurun_out/r2b_bench.err
timeout 900 python -m pytest tests/test_scale_parity.py -x -q > gpurun_out/r2b_scale.log 2>&1
echo "scale rc $?"; tail -30 gpurun_out/r2b_scale.log
