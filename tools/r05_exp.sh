#!/bin/bash
# scratch: per-launch tables of one recogniser pass at 16 / 32 / 64 faces (what the paired / grouped pipeline modes put on the critical path)
set -u
TAG=${1:-r06c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_cpp_shells.py -q -x 2>&1 | tail -3 > "$OUT/${TAG}_pytest.log"
for nf in 16 32 48 64; do NF=$nf bash tools/layer_table.sh "A=1"; done > "$OUT/${TAG}_layers.txt" 2>&1
cat "$OUT/${TAG}_pytest.log"; cat "$OUT/${TAG}_layers.txt" | tr '|' '\n' | awk '{print}' | head -300
