#!/bin/bash
# whole GPU suite + smoke + one short bench without legs
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r3validate_tests.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/r3validate_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/gpu_r3_ab.sh A=1
