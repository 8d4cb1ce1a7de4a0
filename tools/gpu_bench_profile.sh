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
  # the SAME command as the bench line above (defaults: 1e9 rows, 3 steps, 1 warm-up); the CPU baseline
  # leg is skipped under the profiler (it launches nothing on the GPU)
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python $R/bench.py --rows $ROWS --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1; echo "rocprof exit $?"
  f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep "ares::" $f | cut -c1-160 | head -12
  # keep only the summaries (the raw trace can be large)
  find $OUT/prof_$TAG -name '*kernel_trace.csv' -size +8M -delete
fi
if [ "${SKIP_CONFIGS:-0}" != "1" ]; then
  cd $R
  timeout 600 python tools/bench_configs.py c2 c4 hll geo 2>/dev/null | cut -c1-300
fi
