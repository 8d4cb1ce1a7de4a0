#!/bin/bash
# Round 4, race hunt 7: the single-threaded fuzz loop with the harness staging its copies through pinned host memory
# (HostAlloc, as the Go host does) against pageable numpy memory (the HIP runtime stages those copies itself)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r4pinned
mkdir -p $out
N=${1:-16}
pids=()
for i in $(seq 1 $N); do
  for arm in pinned pageable; do
    extra=""; [ $arm = pageable ] && extra="ARES_TEST_PAGEABLE=1"
    ( env ARES_NO_TORCH=1 ARES_TEMP_ORPHANS=1 ARES_FUZZ_DUMP=$out ARES_RTC_CACHE_DIR=/tmp/rtc_${arm}_$i $extra timeout 900 python tools/stress_canary.py \
          --threads none --programs ${PROGRAMS:-640} --tag $arm$i > $out/${arm}_$i.json 2> $out/${arm}_$i.err; echo "$arm$i rc $?" >> $out/rc.txt ) &
    pids+=($!)
  done
done
for p in "${pids[@]}"; do wait $p; done
sort $out/rc.txt | grep -v "rc 0" | tr '\n' ';'
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
cat $out/fuzz_mismatch*.txt 2>/dev/null | head -30
