#!/bin/bash
# scratch: conv_s2_kernel with weight fragments eight steps ahead (two- and four-tile variants): parity, stamps, per-kernel times with the ring on / off
set -u
TAG=${1:-r06h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_gpu_embedder.py tests/test_gpu_headline.py -q -x 2>&1 | tail -4 > "$OUT/${TAG}_pytest.log"
T="FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so"
for nf in 32 64 128; do
  for d in 1 0; do
  echo "== faces $nf deep $d"
  env $T FRT_S2_DEEP=$d FRT_S2_STAMPS=1 python tools/prof_embed.py $nf 20 2>&1 | grep "s2 stamps" | grep first | grep -v "NT 7"
  done
done > "$OUT/${TAG}_s2_stamps.txt" 2>&1
for nf in 32 128; do NF=$nf NROWS=14 bash tools/quick_embed_prof.sh "$T FRT_S2_DEEP=1" "$T FRT_S2_DEEP=0" "$T FRT_S2_DEEP=1" "$T FRT_S2_DEEP=0" | grep -E "==|total|conv_s2_kernel<[24]"; done > "$OUT/${TAG}_embed.txt" 2>&1
cat "$OUT/${TAG}_pytest.log" "$OUT/${TAG}_s2_stamps.txt" "$OUT/${TAG}_embed.txt"
