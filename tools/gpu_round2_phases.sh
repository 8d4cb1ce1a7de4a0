#!/bin/bash
# phase breakdown of the specialised kernels (ARES_HR_PHASES=1) for live-size and 64 Mi-row batches + leg timings without it
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
C="--leg --null-fraction 0.01 --steps 1 --warmup 1"
if [ "${1:-all}" != "big" ]; then
echo "== live leg timing"; timeout 300 python bench.py $C --rows 1000000000 --batch-rows 2097152 2>gpurun_out/ph_live.err | cut -c1-700
echo "== live phases"; ARES_HR_PHASES=1 timeout 300 python bench.py $C --rows 134217728 --batch-rows 2097152 2>&1 >/dev/null | grep phases | tail -4
fi
echo "== big leg timing"; timeout 300 python bench.py $C --rows 1000000000 --batch-rows 67108864 2>gpurun_out/ph_big.err | cut -c1-700
echo "== big phases"; ARES_HR_PHASES=1 timeout 300 python bench.py $C --rows 268435456 --batch-rows 67108864 2>&1 >/dev/null | grep phases | tail -6
