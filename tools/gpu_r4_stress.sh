#!/bin/bash
# Round 4, race hunt 2: the four-thread fuzz round in a loop under library variants (one process each, sequential).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4stress
ITERS=${1:-60}
python -c "import torch" 2>/dev/null
run() {  # tag, env...
  tag=$1; shift
  env "$@" ARES_FILTER_CHECK=gpurun_out/r4stress/fc_$tag.log ARES_FUZZ_DUMP=gpurun_out/r4stress timeout 400 \
    python tools/stress_fuzz.py --iters $ITERS --tag $tag > gpurun_out/r4stress/$tag.json 2> gpurun_out/r4stress/$tag.err
  echo "$tag rc $? $(tail -1 gpurun_out/r4stress/$tag.json | cut -c1-600)"
  grep -h "MISMATCH" gpurun_out/r4stress/fc_$tag.log 2>/dev/null | head -5
  tail -3 gpurun_out/r4stress/$tag.err | cut -c1-300
}
run orphans ARES_TEMP_ORPHANS=1
run control ARES_TEMP_ORPHANS=0
run orphans_destroysync ARES_TEMP_ORPHANS=1 ARES_DESTROY_SYNC=1
run orphans_syncfree ARES_TEMP_ORPHANS=1 ARES_MEM_SYNC_FREE=1
run orphans_nodefer ARES_TEMP_ORPHANS=1 ARES_DEFER=0
run orphans_nopool ARES_TEMP_ORPHANS=1 ARES_MEM_POOL=0
run orphans2 ARES_TEMP_ORPHANS=1
ls gpurun_out/r4stress | head -40
cat gpurun_out/r4stress/*.txt 2>/dev/null | head -40
