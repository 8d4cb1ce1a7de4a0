#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ARES_LEAN_MIN_GROUPS=0 timeout 600 python -m pytest -q -x -m gpu -k "hip or cpp_driver or python_mirror or fused_extension or crowded" tests/test_executor.py::test_c3_shape_matches_oracle tests/test_executor.py::test_native_driver_matches_python_executor tests/test_executor.py::test_pending_transforms_are_consumed_by_hash_reduce tests/test_executor.py::test_fused_extension_matches_unfused_sequence tests/test_executor.py::test_specialised_merge_hands_crowded_partitions_to_the_generic_merge > gpurun_out/r2d_lean.log 2>&1
echo "lean rc $?"; tail -25 gpurun_out/r2d_lean.log
timeout 900 python -m pytest tests/test_scale_parity.py -x -q -k "fused or two_streams or live or extension" > gpurun_out/r2d_scale.log 2>&1
echo "scale rc $?"; tail -15 gpurun_out/r2d_scale.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-legs --no-cpu-baseline > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2d_bench.json'))
print(d['value'], d['ms_per_step'], d['check_groups'])
for k,v in d['kernels'].items(): print(k, v)
PY
tail -5 gpurun_out/r2d_bench.err
