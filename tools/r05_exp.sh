#!/bin/bash
set -u
TAG=${1:-r05h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_gpu_detector.py tests/test_gpu_postproc.py tests/test_gpu_match.py -q 2>&1 | tail -15 > "$OUT/${TAG}_pytest.log"
cd /tmp && export TMPDIR=/tmp
trace() {  # name, env...
  local name=$1; shift
  rm -rf /tmp/prof_det && mkdir -p /tmp/prof_det
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o st -- python "$ROOT/tools/prof_det.py" ${PB:-32} 5 > /dev/null 2>&1
  cp "$(find /tmp/prof_det -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_det_${name}.csv" 2>/dev/null
}
trace b32 X=1
PB=1 trace b1 X=1
cd "$ROOT"
for f in "$OUT"/${TAG}_det_*.csv; do echo "== $f"; python tools/det_table.py "$f"; done > "$OUT/${TAG}_det_tables.txt" 2>&1
for i in 1 2; do
  python bench.py --steps 100 --no-cpu-baseline --no-extras --no-profile > "$OUT/${TAG}_bench_dual1_$i.json" 2>/dev/null
  FRT_PIPELINE_DUAL_EMBED=0 python bench.py --steps 100 --no-cpu-baseline --no-extras --no-profile > "$OUT/${TAG}_bench_dual0_$i.json" 2>/dev/null
done
ls -la "$OUT"
