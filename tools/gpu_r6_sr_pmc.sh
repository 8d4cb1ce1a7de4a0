#!/bin/bash
# PMC passes over the Sort + Reduce merge kernel (C3 dims, COUNT(*)): where its wave cycles go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6/sr_pmc; mkdir -p $OUT
CMD="python $R/bench.py --leg --steps 1 --warmup 0 --rows 2e8 --batch-rows 67108864 --sort-path count"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o t -- $CMD > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][:40]
    if "sr_merge" not in k and "sr_scan" not in k and "hr_scan" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: round(v / max(1, n[(k, c)])) for c, v in d.items()})
PY
done
