#!/bin/bash
# Round 4, race hunt: N concurrent runs of the whole GPU suite with the stream temporaries kept across stream destruction
# (ARES_TEMP_ORPHANS=1), the filter self-check on (ARES_FILTER_CHECK) and the fuzzer's device dump on (ARES_FUZZ_DUMP).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4race
N=${1:-3}
EXTRA="${2:-}"
python -c "import torch" 2>/dev/null   # page the image in once
pids=()
for i in $(seq 1 $N); do
  ( env ARES_TEMP_ORPHANS=1 ARES_FILTER_CHECK=gpurun_out/r4race/fc_$i.log ARES_FUZZ_DUMP=gpurun_out/r4race $EXTRA \
      ARES_RTC_CACHE_DIR=/tmp/rtc_$i timeout 780 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4race/suite_$i.log 2>&1
    echo "suite $i rc $?" >> gpurun_out/r4race/rc.txt ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
cat gpurun_out/r4race/rc.txt
for i in $(seq 1 $N); do tail -3 gpurun_out/r4race/suite_$i.log; grep -c MISMATCH gpurun_out/r4race/fc_$i.log; tail -2 gpurun_out/r4race/fc_$i.log; done
ls gpurun_out/r4race | head -30
cat gpurun_out/r4race/*.txt 2>/dev/null | head -40
