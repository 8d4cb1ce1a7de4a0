#!/bin/bash
# where does a cold process's slow batch spend its time?  ARES_RTC_TRACE: every entry point / launch / libmem call that kept
# the calling thread for more than 5 ms, next to the code-object events
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4
tmp=$(mktemp -d)
ARGS="--leg --cold --rows 1e9 --batch-rows 67108864 --steps 3 --warmup 1"
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  first query", d["cold_first_query_ms"], [round(x, 1) for x in d["cold_first_query_batch_ms"]])
print("  new constants", d.get("new_constants_ms"), [round(x, 1) for x in d.get("new_constants_batch_ms", [])][:4])
PY
}
for run in fill warm1 warm2 warm3; do
  ARES_RTC_TRACE=$R/gpurun_out/r4/ct_$run.trace ARES_RTC_CACHE_DIR=$tmp timeout 300 python bench.py $ARGS > gpurun_out/r4/ct_$run.json 2>gpurun_out/r4/ct_$run.err
  echo "== $run rc $?"; show gpurun_out/r4/ct_$run.json; cat gpurun_out/r4/ct_$run.trace | cut -c1-160
done
echo "== as children of bench.py"
ARES_RTC_TRACE=$R/gpurun_out/r4/ct_bench.trace timeout 600 python bench.py --steps 3 --warmup 1 --legs cold --no-cpu-baseline --no-pmc > gpurun_out/r4/ct_bench.json 2>gpurun_out/r4/ct_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/ct_bench.json").read().strip().splitlines()[-1])
for k, v in d["legs"].items():
    if "cold" in k or "constants" in k or "disk" in k: print(" ", k, v if not isinstance(v, list) else [round(x, 1) for x in v])
PY
cut -c1-160 gpurun_out/r4/ct_bench.trace
timeout 600 python -m pytest tests -m gpu -q -x -k "lookup or multi or transform" 2>&1 | tail -3
