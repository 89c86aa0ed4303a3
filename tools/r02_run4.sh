#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_match.py tests/test_gpu_embedder.py tests/test_gpu_pipeline.py tests/test_gpu_headline.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py --steps 50 --no-cpu-baseline --stage-profile $O/stages.json 2> $O/bench.err | head -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d/bench.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'resident', d['hbm_resident']['ms_per_step'], 'roofline', r['frac'], r['avg_launch_us'], r.get('serial_avg_launch_us'))
print({k:v['avg_launch_us'] for k,v in r['all_3x3_conv_kernels']['per_kernel'].items()})
s=json.load(open('gpurun_out/r02d/stages.json')); print({k:round(v['ms_per_step'],3) for k,v in s.items()})
PY
export FRT_LIB=$GRAFT_REPO_ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so
for NF in 4 16 32 64 128; do
  echo "##### F=$NF"
  NF=$NF NROWS=4 bash tools/quick_embed_prof.sh "FRT_CONV_SMALL_BATCH=0" "FRT_CONV_SMALL_BATCH=1" 2>&1 | grep -E "==|total|conv_patch"
done > $O/small_batch.log 2>&1
cat $O/small_batch.log
