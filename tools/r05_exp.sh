#!/bin/bash
# scratch: prologues without run-time integer divisions (conv_patch_kernel, conv_s2_kernel): parity tests, per-kernel A/B against the round-4 library, bench
set -u
TAG=${1:-r06f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_gpu_embedder.py tests/test_gpu_headline.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -4 > "$OUT/${TAG}_pytest.log"
for nf in 128 32; do echo "=== faces $nf"; NF=$nf NROWS=12 bash tools/quick_embed_prof.sh "X=1" "FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_r04.so FRT_LIB_OLD=1" "X=2" "FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_r04.so FRT_LIB_OLD=1"; done > "$OUT/${TAG}_embed_ab.txt" 2>&1
cd "$ROOT"
for i in 1 2 3; do
for L in new old; do
  if [ $L = old ]; then E="FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_r04.so FRT_LIB_OLD=1"; else E="X=1"; fi
  env $E python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$L', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'])"
done; done > "$OUT/${TAG}_bench_ab.txt" 2>&1
cat "$OUT/${TAG}_pytest.log" "$OUT/${TAG}_embed_ab.txt" "$OUT/${TAG}_bench_ab.txt"
