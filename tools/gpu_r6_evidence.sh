#!/bin/bash
# Round 6 evidence (runs on the GPU box through gpurun): for the headline and for every leg DESIGN.md quotes, a rocprofv3
# --kernel-trace --stats summary and two --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, never combined with API
# tracing), plus two A/B pairs.  Everything lands in gpurun_out/r6ev/; the summaries are copied to profiles/ by hand.
#   bash tools/gpu_r6_evidence.sh [sections]      sections: any of  c3 sort trips archive live c2 c4 ab
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6ev
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
WHAT=${*:-c3 sort trips archive live c2 c4 ab}
trace() {
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o t -- "$@" > $OUT/prof_$tag.log 2>&1
  echo "$tag trace exit $?"
  f=$(find $OUT/prof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv && grep -E "ares::|_rtc" $f | cut -c1-160 | head -8
  find $OUT/prof_$tag -name '*kernel_trace.csv' -delete
}
pmc() {
  tag=$1; rows=$2; shift; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$tag/pass_$c -o p -- "$@" > $OUT/pmc_${tag}_$c.log 2>&1
    echo "$tag pmc $c exit $?"
  done
  python $R/tools/pmc_summary.py $OUT/pmc_$tag $rows > $OUT/${tag}_pmc_summary.md 2>&1
  cp $OUT/pmc_$tag/traffic.json $OUT/${tag}_pmc_traffic.json 2>/dev/null
  find $OUT/pmc_$tag -name '*.csv' -delete
  head -12 $OUT/${tag}_pmc_summary.md | cut -c1-200
}
LIVE="python $R/bench.py --batch-rows 2097152 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-legs --leg"
for w in $WHAT; do
  case $w in
    c3)
      trace c3 python $R/bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline --no-pmc
      pmc c3 67108864 python $R/tools/pmc_driver.py 67108864 3 ;;
    sort)   # the reference's default aggregation path over the C3 dimensions: COUNT(*) through Sort + Reduce (sort_reduce_fused.hip)
      trace sort python $R/bench.py --leg --steps 3 --warmup 1 --rows 1e9 --batch-rows 67108864 --sort-path count
      pmc sort 67108864 python $R/bench.py --leg --steps 2 --warmup 1 --rows 268435456 --batch-rows 67108864 --sort-path count ;;
    archive)
      trace archive python $R/bench.py --leg --steps 3 --warmup 1 --rows 1e9 --batch-rows 67108864 --archive --ts-range 3600,601200 ;;
    c3t) trace c3 python $R/bench.py --steps 5 --warmup 2 --no-legs --no-cpu-baseline --no-pmc ;;   # (traces only: no counter passes)
    livet) trace live $LIVE --rows 1e9 ;;
    live)
      trace live $LIVE --rows 1e9
      pmc live 2097152 $LIVE --rows 2e8 ;;
    trips)
      trace trips env TRIPS_ROWS=1e9 python $R/tools/bench_configs.py trips
      pmc trips 67108864 env TRIPS_ROWS=268435456 python $R/tools/bench_configs.py trips ;;
    c2)
      trace c2 python $R/tools/bench_configs.py c2
      pmc c2 100000000 python $R/tools/bench_configs.py c2 ;;
    c4)
      trace c4 env C4_ROWS=1e9 python $R/tools/bench_configs.py c4spec
      pmc c4 67108864 env C4_ROWS=268435456 C4_KEYS=5e7 python $R/tools/bench_configs.py c4spec ;;
    ab)   # what round 6 added, switched off and on again (same box, back to back)
      for v in 1 0; do
        ARES_SORT_VECTORS=$v python $R/tools/bench_configs.py c4spec 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=d[0] if isinstance(d,list) else d; print('c4 1B rows / 50M keys, ARES_SORT_VECTORS=$v:', round(d['ms'],1), 'ms', d['key_level_check'], {k: round(x['avg_ms'],3) for k,x in list(d['kernels'].items())[:6]})"
      done
      for v in 1 0; do
        ARES_SORT_FUSE=$v python $R/bench.py --leg --steps 3 --warmup 1 --rows 1e9 --batch-rows 67108864 --sort-path count 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 COUNT(*) through Sort + Reduce, ARES_SORT_FUSE=$v:', round(d['ms_per_step'],2), 'ms per 1 B rows', d['check_groups'])"
        ARES_EXPAND_RLE=$v python $R/bench.py --leg --steps 3 --warmup 2 --rows 1e9 --batch-rows 67108864 --archive --ts-range 3600,601200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('archive batches, ARES_EXPAND_RLE=$v:', round(d['ms_per_step'],2), 'ms per 1 B rows', d['check_groups'])"
        ARES_SORT_STATE=$v python $R/bench.py --leg --steps 3 --warmup 1 --rows 1e9 --batch-rows 67108864 --sort-path count 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 COUNT(*) through Sort + Reduce, ARES_SORT_STATE=$v:', round(d['ms_per_step'],2), 'ms per 1 B rows', d['check_groups'])"
      done ;;
  esac
done
ls $OUT | head -60
