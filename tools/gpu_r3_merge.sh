#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_executor.py tests/test_scale_parity.py tests/test_sequence_fuzz.py tests/test_hip_parity.py -m gpu -q -x > gpurun_out/r3i_tests.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r3i_tests.log
bash tools/gpu_r3_ab.sh X=1 X=2
