#!/bin/bash
# suite + bench (selected legs) + C2 at its stated size
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r3g}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/suite_$TAG.log 2>&1
echo "pytest rc $?" >> gpurun_out/suite_$TAG.log
grep -E "^FAILED|^ERROR|passed|failed|rc " gpurun_out/suite_$TAG.log | head -20
timeout 300 python tools/bench_configs.py c2 > gpurun_out/c2_$TAG.json 2> gpurun_out/c2_$TAG.err; echo "c2 rc $?"; cut -c1-700 gpurun_out/c2_$TAG.json
BENCH_ARGS="${BENCH_ARGS:---legs groups --no-cpu-baseline}" bash tools/gpu_r3_bench.sh $TAG
