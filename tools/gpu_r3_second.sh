#!/bin/bash
# the GPU suite with libmem.so verifying every "still cleared" block it hands out; then the fuzz chunk that failed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ARES_MEM_VERIFY_CLEAN=1 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_write_tracking.py > gpurun_out/r3b_suite_verify.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3b_suite_verify.log
grep -n "libmem:" gpurun_out/r3b_suite_verify.log | head -5
tail -15 gpurun_out/r3b_suite_verify.log
for i in 1 2 3; do timeout 300 python -m pytest tests/test_sequence_fuzz.py -m gpu -q -x 2>&1 | tail -3; done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r3b_bench20.json 2> gpurun_out/r3b_bench20.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3b_bench20.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step','max_ms_per_step')})
print(d['config']['host_ms_of_each_step'], d['config']['profiled_pass_ms_per_step'])
PY
