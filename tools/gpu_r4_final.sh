#!/bin/bash
# Round-4 closing measurements on the GPU box: the driver's bench command (all legs, live PMC traffic, CPU baseline),
# then rocprofv3 --kernel-trace --stats of the same command without legs.  Outputs in gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r4final}
mkdir -p $OUT
cd $R
for i in $(seq 1 ${SUITES:-0}); do timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/suite_${TAG}_$i.log 2>&1; echo "suite $i rc $? $(tail -1 $OUT/suite_${TAG}_$i.log)"; done
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --leg-budget 0 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench exit $? after $(( $(date +%s) - t0 )) s"
tail -3 $OUT/bench_$TAG.err
python - <<PY
import json
d=json.loads([l for l in open('$OUT/bench_$TAG.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step','max_ms_per_step')}, d['check_groups']['status'])
print('roofline', {k:v for k,v in d['roofline'].items() if k!='note'})
print('cpu', d.get('cpu_baseline'))
for k,v in d['legs'].items():
    if not isinstance(v, (dict, list)): print(k, v); continue
    if isinstance(v, list):
        for row in v: print(k, json.dumps({a:b for a,b in row.items() if a!='kernels'})[:300])
        continue
    v=dict(v); v.pop('kernels',None); v.pop('roofline',None)
    print(k, json.dumps(v)[:400])
PY
export TMPDIR=/tmp
cd /tmp
t0=$(date +%s)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-pmc > $OUT/prof_$TAG.log 2>&1; echo "rocprof exit $? after $(( $(date +%s) - t0 )) s"
f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 $f | cut -c1-220
find $OUT/prof_$TAG -name '*kernel_trace.csv' -delete
