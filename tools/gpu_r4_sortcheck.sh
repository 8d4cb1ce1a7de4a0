#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_golden_vectors.py -m gpu -q -p no:cacheprovider -k "sort or Sort or reduce or hyperloglog" 2>&1 | tail -5
bash tools/gpu_r4_c4ab.sh X=1 ARES_SORT_TOPBITS=0
