#!/bin/bash
# (1) the sequence fuzzer repeated with stderr visible (pytest -s) and the run-time compiled kernels off: what aborts?
#     A/B on ARES_TEMP_ORPHANS;  (2) the whole suite on the new run-time compiled kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for i in $(seq 1 12); do
  ARES_RTC=0 timeout 300 python -X faulthandler -m pytest tests/test_sequence_fuzz.py -m gpu -q -x -s > gpurun_out/r3d_new_$i.log 2>&1
  rc=$?; echo "orphans on $i rc $rc $(tail -1 gpurun_out/r3d_new_$i.log | cut -c1-80)"
  [ $rc -ne 0 ] && grep -v "^  File\|^Thread\|^$" gpurun_out/r3d_new_$i.log | head -12
done
for i in $(seq 1 12); do
  ARES_RTC=0 ARES_TEMP_ORPHANS=0 timeout 300 python -X faulthandler -m pytest tests/test_sequence_fuzz.py -m gpu -q -x -s > gpurun_out/r3d_off_$i.log 2>&1
  rc=$?; echo "orphans off $i rc $rc $(tail -1 gpurun_out/r3d_off_$i.log | cut -c1-80)"
  [ $rc -ne 0 ] && grep -v "^  File\|^Thread\|^$" gpurun_out/r3d_off_$i.log | head -12
done
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3d_suite.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3d_suite.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3d_suite.log | head -30
