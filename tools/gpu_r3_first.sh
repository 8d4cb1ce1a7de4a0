#!/bin/bash
# round 3, first GPU call: the whole suite (new: global-table HashReduce cases re-armed, write tracking under
# ARES_MEM_VERIFY_CLEAN=1) + the driver's bench command line with the allocator's debug counters on
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_suite.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3a_suite.log
tail -8 gpurun_out/r3a_suite.log
ARES_MEM_DEBUG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r3a_bench20.json 2> gpurun_out/r3a_bench20.err
echo "bench rc $?"; tail -5 gpurun_out/r3a_bench20.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3a_bench20.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step','max_ms_per_step')})
print(d['config']['host_ms_of_each_step'], d['config']['libmem_driver_calls_in_timed_region'], d['config']['profiled_pass_ms_per_step'])
print(d['roofline'])
print({k:(round(v['avg_ms'],4),v['launches']) for k,v in d['kernels'].items()})
PY
