#!/bin/bash
set -u
TAG=${1:-r05e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_gpu_embedder.py tests/test_gpu_headline.py -q 2>&1 | tail -25 > "$OUT/${TAG}_pytest_embed.log"
python tools/dynamic_range_sweep.py --out "$OUT/${TAG}_dynamic_range.json" 2>&1 | grep -v amdgpu.ids | grep branch > "$OUT/${TAG}_dynamic_range.txt"
ls -la "$OUT"
