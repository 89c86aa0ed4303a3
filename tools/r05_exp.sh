#!/bin/bash
# scratch: validation of the recogniser epilogue change (prefetched registers landed before the first store)
set -u
TAG=${1:-r05t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_gpu_embedder.py tests/test_gpu_headline.py -q -x 2>&1 | tail -6 > "$OUT/${TAG}_pytest.log"
NROWS=16 bash tools/quick_embed_prof.sh "X=1" "FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_r04.so FRT_LIB_OLD=1" > "$OUT/${TAG}_embed_ab.txt" 2>&1
cd "$ROOT"
for i in 1 2; do
python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_$i.json"
python - "$OUT/${TAG}_bench_$i.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('sustained_peak',{}).get('achieved_over_sustained'))
PY
done
cat "$OUT/${TAG}_pytest.log" "$OUT/${TAG}_embed_ab.txt"
