#!/bin/bash
# Round 4, race hunt 6: the single-threaded fuzz loop with and without torch (and its bundled ROCm runtime) in the process
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r4torch
mkdir -p $out
N=${1:-16}
python -c "import torch" 2>/dev/null
pids=()
for i in $(seq 1 $N); do
  ( env ARES_NO_TORCH=1 ARES_TEMP_ORPHANS=1 ARES_FUZZ_DUMP=$out ARES_RTC_CACHE_DIR=/tmp/rtc_nt_$i timeout 800 python tools/stress_canary.py --threads none \
        --programs ${PROGRAMS:-320} --tag notorch$i > $out/notorch_$i.json 2> $out/notorch_$i.err; echo "notorch$i rc $?" >> $out/rc.txt ) &
  pids+=($!)
  ( env ARES_TEMP_ORPHANS=1 ARES_FUZZ_DUMP=$out ARES_RTC_CACHE_DIR=/tmp/rtc_t_$i timeout 800 python tools/stress_canary.py --threads none \
        --programs ${PROGRAMS:-320} --tag torch$i > $out/torch_$i.json 2> $out/torch_$i.err; echo "torch$i rc $?" >> $out/rc.txt ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
sort $out/rc.txt | tr '\n' ';'
echo
cat $out/*.json | python -c "
import sys, json
agg = {}
for line in sys.stdin:
    try: d = json.loads(line)
    except ValueError: continue
    k = d['tag'].rstrip('0123456789')
    a = agg.setdefault(k, {'procs': 0, 'programs': 0, 'mismatches': 0, 'hits': 0, 'first': []})
    a['procs'] += 1; a['programs'] += d['programs']; a['mismatches'] += d['fuzz_mismatches']; a['hits'] += d['canary_hits']; a['first'] += d['bad'][:2] + d['hits'][:2]
print(json.dumps(agg))
"
grep -c . $out/*.txt 2>/dev/null | head; cat $out/fuzz_mismatch*.txt 2>/dev/null | head -30
python -c "
import sys; sys.path.insert(0, 'tests')
import os, ctypes
import torch
print('torch', torch.__version__, torch.version.hip)
import harness as H
H.hip_backend()
print([l.split()[-1] for l in open('/proc/self/maps') if any(s in l for s in ('amdhip', 'hsa-runtime', 'hiprtc', 'comgr')) and 'r-xp' in l])
"
