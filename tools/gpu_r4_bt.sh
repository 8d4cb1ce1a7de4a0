#!/bin/bash
# Round 4, race hunt 8: where do the rare aborts / segmentation faults of the fuzz loop come from?  Native backtraces
# (ARES_BACKTRACE=1) of many short single-threaded loops; pinned staging in the harness (the default).
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r4bt
mkdir -p $out
N=${1:-40}
T=${2:-240}
pids=()
for i in $(seq 1 $N); do
  ( env ARES_NO_TORCH=1 ARES_BACKTRACE=1 ARES_RTC_CACHE_DIR=/tmp/rtc_bt_$i ${EXTRA_ENV} timeout $T python -X faulthandler tools/stress_canary.py --threads none \
        --programs ${PROGRAMS:-640} --tag bt$i > $out/bt_$i.json 2> $out/bt_$i.err; echo "bt$i rc $?" >> $out/rc.txt ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
sort $out/rc.txt | grep -v "rc 0" | tr '\n' ';'
echo
cat $out/*.json | python -c "
import sys, json
n = p = m = 0
for line in sys.stdin:
    try: d = json.loads(line)
    except ValueError: continue
    n += 1; p += d['programs']; m += d['fuzz_mismatches']
print({'finished': n, 'programs': p, 'mismatches': m})
"
for f in $out/*.err; do if grep -q "native stack\|terminate\|Fatal" $f; then echo "== $f"; grep -v "amdgpu.ids" $f | head -60 | cut -c1-220; fi; done | head -250
