#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ubench_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $pmc | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/$tag -o p -- $R/tools/bin/ubench_scatter > $OUT/$tag.log 2>&1
  echo "$pmc exit $?"
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/ubench_pmc'
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:110]][r['Counter_Name']].append(float(r['Counter_Value']))
names=sorted({c for k in acc.values() for c in k})
with open(root+'/summary.txt','w') as o:
    o.write('kernel | '+' | '.join(names)+'\n')
    for k in acc:
        o.write(k+' | '+' | '.join(('%.4g'%(sum(acc[k][c])/len(acc[k][c])) if c in acc[k] else '-') for c in names)+'\n')
print(open(root+'/summary.txt').read())
PY
find $OUT -name '*kernel_trace.csv' -delete
