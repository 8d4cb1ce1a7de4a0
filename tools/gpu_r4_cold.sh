#!/bin/bash
# cold-start legs with the RTC build trace: empty disk cache, then a second process that finds the code objects on disk
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
python -c "import torch" 2>/dev/null
tmp=$(mktemp -d)
for i in 1 2; do
  rm -f gpurun_out/r4/rtc_trace_$i.log
  ARES_RTC_CACHE_DIR=$tmp ARES_RTC_TRACE=gpurun_out/r4/rtc_trace_$i.log timeout 300 python bench.py --leg --cold --rows 1e9 --batch-rows 67108864 --steps 3 --warmup 1 ${BENCH_ARGS} > gpurun_out/r4/cold_$i.json 2> gpurun_out/r4/cold_$i.err
  echo "cold $i rc $?"; cut -c1-1400 gpurun_out/r4/cold_$i.json; echo; cat gpurun_out/r4/rtc_trace_$i.log
done
