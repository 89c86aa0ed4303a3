#!/bin/bash
# scratch: conv3x3_split - output channels split over two workgroups at a few frames per call; second workgroup of a CU started late (skew)
set -u
TAG=${1:-r06a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
{
for b in 1 4 32; do echo "== frames $b"; timeout 300 tools/ubench/det_conv3h_bench $b | grep -v "^ssh 16"; done
for k in 1 2 3 5; do echo "== skew $k, frames 32"; timeout 300 tools/ubench/det_conv3h_bench_skew$k 32 | grep -v "^ssh 16"; done
} > "$OUT/${TAG}_conv3h.txt" 2>&1
python -m pytest tests/test_gpu_detector.py tests/test_gpu_headline.py -q -x 2>&1 | tail -5 > "$OUT/${TAG}_pytest.log"
cd /tmp && export TMPDIR=/tmp
trace() {  # name, env...
  local name=$1; shift
  rm -rf /tmp/prof_det && mkdir -p /tmp/prof_det
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o st -- python "$ROOT/tools/prof_det.py" ${PB:-32} 5 > /dev/null 2>&1
  cp "$(find /tmp/prof_det -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_det_${name}.csv" 2>/dev/null
}
T="FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so"
for b in 1 2 4 8; do
  PB=$b trace split_b$b $T
  PB=$b trace nosplit_b$b $T FRT_C3H_SPLIT_TILES=0
done
cd "$ROOT"
for f in "$OUT"/${TAG}_det_*.csv; do echo "== $(basename $f)"; python tools/det_table.py "$f" | grep -E "conv3x3_split|kernels per"; done > "$OUT/${TAG}_det_tables.txt" 2>&1
cat "$OUT/${TAG}_conv3h.txt" "$OUT/${TAG}_det_tables.txt" "$OUT/${TAG}_pytest.log"
