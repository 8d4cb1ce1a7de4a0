#!/bin/bash
# Round 4, race hunt 3: which earlier test file leaves the state behind that makes the fuzzer fail?  One process per
# predecessor file (file + the fuzz file three times), plus N whole-suite runs, all concurrently, all with
# ARES_TEMP_ORPHANS=1 and the self-checks on.
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r4bisect
mkdir -p $out
NSUITE=${1:-3}
python -c "import torch" 2>/dev/null
pids=()
launch() {  # tag, pytest args...
  tag=$1; shift
  ( env ARES_TEMP_ORPHANS=1 ARES_FILTER_CHECK=$out/fc_$tag.log ARES_FUZZ_DUMP=$out ARES_RTC_CACHE_DIR=/tmp/rtc_$tag \
      timeout 840 python -m pytest "$@" -m gpu -q -p no:cacheprovider --keep-duplicates > $out/$tag.log 2>&1
    echo "$tag rc $? $(tail -1 $out/$tag.log | cut -c1-100)" >> $out/rc.txt ) &
  pids+=($!)
}
F=tests/test_sequence_fuzz.py
for f in test_1k_trips test_arrays test_baseline_configs test_bench_distributed test_executor test_expand_libmem test_golden_vectors \
         test_hip_parity test_host_batches test_scale_parity test_shard_merge test_write_tracking test_prototypes; do
  [ -f tests/$f.py ] && launch $f tests/$f.py $F $F $F
done
for i in $(seq 1 $NSUITE); do launch suite$i tests; done
for p in "${pids[@]}"; do wait $p; done
sort $out/rc.txt
grep -l "MISMATCH" $out/fc_*.log 2>/dev/null
grep -h "MISMATCH" $out/fc_*.log 2>/dev/null | cut -c1-700 | head -20
cat $out/*.txt 2>/dev/null | grep -v " rc " | head -30
for f in $out/*.log; do grep -H "^FAILED\|^ERROR\|AssertionError: (" $f | head -5; done
