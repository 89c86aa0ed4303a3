#!/bin/bash
# last checks of a round on the GPU box: smoke(), the concurrency-sensitive suites three times, the driver's bench command three times
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
{ python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  for i in 1 2 3; do
    python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_coalesce.py tests/test_gpu_headline.py tests/test_gpu_match.py -m gpu -q 2>&1 | tail -1
  done
  for i in 1 2 3; do
    python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver command', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
} > "$OUT/${TAG}_final_checks.txt" 2>&1
cat "$OUT/${TAG}_final_checks.txt"
