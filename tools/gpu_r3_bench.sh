#!/bin/bash
# bench with all legs (driver command line), then the fuzz loop with C-level stderr visible
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-r3e}
timeout 1500 python bench.py --gpus 1 ${BENCH_STEPS:---steps 20 --warmup 5} $BENCH_ARGS > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc $?"; tail -5 gpurun_out/bench_$TAG.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/bench_$TAG.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step','max_ms_per_step')})
print(d['config']['rtc_kernels_after_priming_pass'], d['config']['profiled_pass_ms_per_step'], d['check_groups']['status'])
print('roofline', d['roofline'])
print('all', d['roofline_all_kernels'])
print({k:(round(v['avg_ms'],4),v['launches'], round(v.get('hbm_GBps',0))) for k,v in d['kernels'].items()})
for k,v in d['legs'].items():
    if not isinstance(v, (dict, list)): print(k, v); continue
    if isinstance(v, list):
        for row in v: print(k, json.dumps(row)[:500])
        continue
    v=dict(v); ks=v.pop('kernels',None)
    print(k, json.dumps(v)[:600])
    if ks: print('    ', {n:(round(x['avg_ms'],4), x['launches']) for n,x in list(ks.items())[:5]})
PY
for i in $(seq 1 ${FUZZ_LOOPS:-0}); do
  timeout 300 python -X faulthandler -m pytest tests/test_sequence_fuzz.py -m gpu -q -x --capture=sys > gpurun_out/r3e_fuzz_$i.log 2>&1
  rc=$?; echo "fuzz $i rc $rc $(tail -1 gpurun_out/r3e_fuzz_$i.log | cut -c1-80)"
  [ $rc -ne 0 ] && grep -v "^  File\|^Thread\|^$" gpurun_out/r3e_fuzz_$i.log | head -12
done
