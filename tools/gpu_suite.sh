#!/bin/bash
# the whole GPU suite (as the driver runs it) + smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/suite_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/suite_pytest.log
tail -25 gpurun_out/suite_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
