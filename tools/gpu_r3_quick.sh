#!/bin/bash
# quick GPU check of a kernel change: A/B bench runs for the env settings before `--`, then pytest -m gpu on the paths after
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cfgs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done; shift
[ ${#cfgs[@]} -gt 0 ] && bash tools/gpu_r3_ab.sh "${cfgs[@]}"
if [ $# -gt 0 ]; then
  timeout 900 python -m pytest "$@" -m gpu -q -x > gpurun_out/quick_tests.log 2>&1
  echo "pytest rc $?"; tail -5 gpurun_out/quick_tests.log
fi
