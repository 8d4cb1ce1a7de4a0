#!/bin/bash
# cold first query of the first GPU process on a fresh box (BootstrapDevice warms device memory), then BootstrapDevice's own cost
d=$(mktemp -d)
ARES_RTC_CACHE_DIR=$d timeout 300 python bench.py --leg --cold --rows 1e9 --batch-rows 67108864 --steps 3 --warmup 1 2>/dev/null | tail -1 > /tmp/cold.json
python3 -c "
import json; d=json.load(open('/tmp/cold.json')); print({k:d[k] for k in ('cold_first_query_ms','cold_first_query_batch_ms','warm_query_ms','cold_check_groups')})"
python3 - <<'PY'
import time,sys
sys.path.insert(0,".")
from aresdb_amd import abi
be=abi.load_hip_backend()
for i in range(3):
    t=time.perf_counter(); be.call("BootstrapDevice"); print("BootstrapDevice", round((time.perf_counter()-t)*1e3,1),"ms")
PY
