#!/bin/bash
set -u
TAG=${1:-r05i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
for B in 32 16 8 64; do
  python bench.py --batch $B --steps $((3200 / B)) --no-cpu-baseline --no-extras --no-profile > "$OUT/${TAG}_bench_b${B}_$i.json" 2>/dev/null
done
done
for f in "$OUT"/*.json; do python -c "
import json
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done > "$OUT/${TAG}_summary.txt"
cat "$OUT/${TAG}_summary.txt"
