#!/bin/bash
# round-end check as the driver runs it: the GPU suite, smoke(), then the default bench line (N = 1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/final_pytest.log
tail -4 gpurun_out/final_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"].get("traffic"))
    print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("legs", {}).items()})
except Exception as e:
    print("bench line unreadable:", e)
PY
