#!/bin/bash
# rocprofv3 counter passes (each in its own run, kernel-trace only — never combined with sys/hip
# tracing) over tools/pmc_driver.py.  Output: gpurun_out/pmc_<tag>/<pass>/..._counter_collection.csv
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${1:-r1}
ROWS=${2:-67108864}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pass$i -o p -- python $R/tools/pmc_driver.py $ROWS 2 > $OUT/pass$i.log 2>&1
  echo "pass $i ($pmc) exit $?"
done
python $R/tools/pmc_summary.py $OUT $ROWS > $OUT/summary.md 2>&1
cat $OUT/summary.md
find $OUT -name '*kernel_trace.csv' -delete
