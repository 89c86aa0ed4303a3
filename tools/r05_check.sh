#!/bin/bash
# round-5 verification of the tree on the GPU box: LDS-DMA RAW probe, GPU suite, driver bench line, detector trace at 32 frames
set -u
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
[ -x tools/ubench/det_conv3h_bench ] && { timeout 300 tools/ubench/det_conv3h_bench 32; timeout 120 tools/ubench/det_conv3h_bench 4; timeout 120 tools/ubench/det_conv3h_bench 1; } > "$OUT/${TAG}_det_conv3h_bench.txt" 2>&1
[ -x tools/ubench/lds_dma_raw ] && timeout 120 tools/ubench/lds_dma_raw > "$OUT/${TAG}_lds_dma_raw.txt" 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > "$OUT/${TAG}_pytest_gpu.log"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver.json" 2> "$OUT/${TAG}_bench_driver.stderr"
cd /tmp && export TMPDIR=/tmp
for B in 32; do
  rm -rf /tmp/prof_det && mkdir -p /tmp/prof_det
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o st -- python "$ROOT/tools/prof_det.py" $B 5 > /dev/null 2>&1
  cp "$(find /tmp/prof_det -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_det_kernel_stats_b$B.csv" 2>/dev/null
done
cd "$ROOT"
python -c "import __graft_entry__ as e; e.smoke()" > "$OUT/${TAG}_smoke.txt" 2>&1
ls -la "$OUT"
