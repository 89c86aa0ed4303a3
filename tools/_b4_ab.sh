#!/bin/bash
# 4 frames per step (BASELINE configs[3] per rank): batches in flight and pipeline switches, alternating runs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r04z; mkdir -p $OUT
cd $ROOT
run() { echo -n "$1 | "; env $2 python bench.py --batch 4 --steps 400 --no-cpu-baseline --no-extras --no-profile $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{ for rep in 1 2; do
  run "in-flight 3 (default)" "A=1" ""
  run "in-flight 4" "A=1" "--in-flight 4"
  run "in-flight 2" "A=1" "--in-flight 2"
  run "one recogniser pass in flight (FRT_PIPELINE_DUAL_EMBED=0)" "FRT_PIPELINE_DUAL_EMBED=0" ""
  run "stage streams at normal priority" "FRT_PIPELINE_STREAM_PRIO=0" ""
  run "hipGraph replay (FRT_PIPELINE_GRAPH=1)" "FRT_PIPELINE_GRAPH=1" ""
done; } > $OUT/r04z_b4_ab.txt 2>&1
cat $OUT/r04z_b4_ab.txt
