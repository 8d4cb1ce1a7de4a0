#!/bin/bash
# suite + bench (selected legs) + phase split of the generated kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r3g}
if [ "${SKIP_SUITE:-0}" != "1" ]; then
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/suite_$TAG.log 2>&1
echo "pytest rc $?" >> gpurun_out/suite_$TAG.log
grep -E "^FAILED|^ERROR|passed|failed|rc " gpurun_out/suite_$TAG.log | head -20
fi
BENCH_ARGS="${BENCH_ARGS:---legs groups,c2 --no-cpu-baseline}" bash tools/gpu_r3_bench.sh $TAG
ARES_HR_PHASES=1 ARES_RTC_ASYNC=0 timeout 300 python tools/pmc_driver.py 67108864 4 2>&1 | grep -E "phases|groups" | tail -6
