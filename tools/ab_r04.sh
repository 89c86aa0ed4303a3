#!/bin/bash
# Same-box A/B of the round-4 final library (built from commit ae5fbac into libfrt_r04.so) against the current one: alternating legs of the
# benchmark's timed region (no side measurements), headline / K = 1 / 4 frames per step / one frame per step.
set -u
TAG=${1:-r05q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
OLD="FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_r04.so FRT_LIB_OLD=1"
run() {  # name, args...
  local name=$1; shift
  for i in 1 2 3; do
    env $OLD python bench.py "$@" --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name r04  ', d['value'], d['ms_per_step'])"
    python bench.py "$@" --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name r05  ', d['value'], d['ms_per_step'])"
  done
}
{
run headline --steps 200
run k1 --faces 1 --steps 200
run b4 --batch 4 --steps 600
run b1 --batch 1 --gallery 10000 --steps 600
} > "$OUT/${TAG}_ab_r04_vs_r05.txt" 2>&1
cat "$OUT/${TAG}_ab_r04_vs_r05.txt"
