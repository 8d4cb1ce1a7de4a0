#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_executor.py tests/test_sequence_fuzz.py tests/test_baseline_configs.py tests/test_host_batches.py -m gpu -q -x > gpurun_out/r3j_tests.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r3j_tests.log
BENCH_ARGS="--legs groups --no-cpu-baseline --no-pmc --steps 6 --warmup 2" bash tools/gpu_r3_bench.sh r3j 2>&1 | grep -E "^groups|value|hr_" | cut -c1-330
