#!/bin/bash
# first GPU pass of round 2: existing parity suite on the new HashReduce kernels, then a short bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_scale_parity.py > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r2a_pytest.log
tail -30 gpurun_out/r2a_pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-legs --no-cpu-baseline > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc $?"
tail -c 3000 gpurun_out/r2a_bench.json
tail -20 gpurun_out/r2a_bench.err
