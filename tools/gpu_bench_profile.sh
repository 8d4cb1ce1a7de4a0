#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, the contract bench, and a rocprofv3
# kernel-trace of a shorter bench run.  Outputs land in gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r1}
ROWS=${2:-1e9}
mkdir -p $OUT
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest exit $?"
  tail -3 $OUT/pytest_gpu_$TAG.log
fi
timeout 1200 python bench.py --rows $ROWS > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench exit $?"
tail -c 3000 $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  export TMPDIR=/tmp
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python $R/bench.py --rows 2.5e8 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1; echo "rocprof exit $?"
  find $OUT/prof_$TAG -name '*kernel_stats*' | head; 
  f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -20 $f
  # keep only the summaries (the raw trace can be large)
  find $OUT/prof_$TAG -name '*kernel_trace.csv' -size +8M -delete
fi
