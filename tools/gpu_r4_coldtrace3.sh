#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4
timeout 120 tools/bin/ubench_malloc 2>&1 | tee gpurun_out/r4/ubench_malloc.txt
bash tools/gpu_r4_coldtrace2.sh
timeout 900 python -m pytest tests -m gpu -q -x -k "hash_reduce or fused or scale or baseline or shard" 2>&1 | tail -3
