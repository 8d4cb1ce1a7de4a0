#!/bin/bash
# Round 4, race hunt 4: host-memory canaries after a thread phase (tools/stress_canary.py); one process per variant, concurrent
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r4canary
mkdir -p $out
python -c "import torch" 2>/dev/null
pids=()
run() {  # tag threads-kind copies env...
  tag=$1; kind=$2; shift 2
  ( env ARES_TEMP_ORPHANS=1 ARES_FUZZ_DUMP=$out ARES_RTC_CACHE_DIR=/tmp/rtc_$tag "$@" timeout 800 python tools/stress_canary.py --threads $kind --tag $tag ${PROGRAMS:+--programs $PROGRAMS} \
      > $out/$tag.json 2> $out/$tag.err; echo "$tag rc $?" >> $out/rc.txt ) &
  pids+=($!)
}
for i in 1 2 3; do run fuzz$i fuzz; done
for i in 1 2; do run none$i none; done
for i in 1 2; do run libmem$i libmem; done
for i in 1 2; do run hip$i hip; done
for i in 1 2; do run hipnull$i hipnull; done
for i in 1 2; do run fuzzdefault$i fuzz ARES_TEMP_ORPHANS=0; done
for p in "${pids[@]}"; do wait $p; done
sort $out/rc.txt
for f in $out/*.json; do cut -c1-900 $f; done
tail -n 3 $out/*.err | cut -c1-300 | grep -v "amdgpu.ids" | head -40
