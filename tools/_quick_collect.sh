#!/bin/bash
# quick verification of the tree: GPU suite, driver bench, one-frame shapes, detector traces at 1 / 32 frames
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > "$OUT/${TAG}_pytest_gpu.log"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver.json" 2> "$OUT/${TAG}_bench_driver.stderr"
python bench.py --batch 1 --gallery 10000 --no-cpu-baseline > "$OUT/${TAG}_bench_config1.json" 2>/dev/null
python bench.py --faces 1 --no-cpu-baseline > "$OUT/${TAG}_bench_k1.json" 2>/dev/null
python bench.py --batch 4 --no-cpu-baseline --steps 300 > "$OUT/${TAG}_bench_b4.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for B in 32 1; do
  rm -rf /tmp/prof_det && mkdir -p /tmp/prof_det
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o st -- python "$ROOT/tools/prof_det.py" $B 5 > /dev/null 2>&1
  cp "$(find /tmp/prof_det -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_det_kernel_stats_b$B.csv" 2>/dev/null
done
ls -la "$OUT"
