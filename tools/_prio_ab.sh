#!/bin/bash
# stage streams prioritised (default) against normal priority (FRT_PIPELINE_STREAM_PRIO=0): headline, K = 1, one frame; alternating runs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r04z; mkdir -p $OUT
cd $ROOT
run() { echo -n "$1 | "; env $2 python bench.py --steps ${4:-300} --no-cpu-baseline --no-extras --no-profile $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{ for rep in 1 2 3; do
  run "headline, prioritised (default)" "A=1" ""
  run "headline, normal priority" "FRT_PIPELINE_STREAM_PRIO=0" ""
done
for rep in 1 2; do
  run "K = 1, prioritised" "A=1" "--faces 1"
  run "K = 1, normal priority" "FRT_PIPELINE_STREAM_PRIO=0" "--faces 1"
  run "one frame / 10k gallery, prioritised" "A=1" "--batch 1 --gallery 10000" 1000
  run "one frame / 10k gallery, normal priority" "FRT_PIPELINE_STREAM_PRIO=0" "--batch 1 --gallery 10000" 1000
done; } > $OUT/r04z_prio_ab.txt 2>&1
cat $OUT/r04z_prio_ab.txt
