#!/bin/bash
# Round 4, race hunt 5: the fuzz loop on the AddressSanitizer build of the host code (make asan), no torch in the process
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r4asan
mkdir -p $out
ASAN_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
N=${1:-6}
pids=()
for i in $(seq 1 $N); do
  kind=none; [ $((i % 2)) -eq 0 ] && kind=fuzz
  ( env LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:log_path=$out/asan_$i \
        ARES_LIB_DIR=$PWD/aresdb_amd/lib_asan ARES_NO_TORCH=1 ARES_TEMP_ORPHANS=1 ARES_RTC_CACHE_DIR=/tmp/rtc_asan_$i \
        timeout 800 python tools/stress_canary.py --threads $kind --programs ${PROGRAMS:-320} --tag asan$i > $out/run_$i.json 2> $out/run_$i.err
    echo "asan$i ($kind) rc $?" >> $out/rc.txt ) &
  pids+=($!)
done
# the same loop on the ordinary build without torch in the process: does torch matter?
for i in 1 2 3; do
  ( env ARES_NO_TORCH=1 ARES_TEMP_ORPHANS=1 ARES_RTC_CACHE_DIR=/tmp/rtc_nt_$i timeout 800 python tools/stress_canary.py --threads none --programs ${PROGRAMS:-320} \
        --tag notorch$i > $out/notorch_$i.json 2> $out/notorch_$i.err; echo "notorch$i rc $?" >> $out/rc.txt ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
sort $out/rc.txt
for f in $out/*.json; do cut -c1-700 $f; done
ls $out | grep asan_ | head
for f in $out/asan_*; do echo "== $f"; head -60 $f | cut -c1-250; done 2>/dev/null | head -150
tail -n 4 $out/*.err | cut -c1-300 | grep -v "amdgpu.ids\|^$" | head -40
