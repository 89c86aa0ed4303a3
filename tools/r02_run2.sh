#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_pipeline.py tests/test_cpp_shells.py -x -q -m gpu > $O/pytest.log 2>&1; tail -12 $O/pytest.log
export FRT_LIB=$GRAFT_REPO_ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so
NF=128 bash tools/quick_embed_prof.sh "X=1" "FRT_CONV_NT4_14=1" > $O/embed_prof.log 2>&1; cat $O/embed_prof.log
unset FRT_LIB
FRT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --no-cpu-baseline 2> $O/bench_dist1.err | head -1 > $O/bench_dist1.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02b/bench_dist1.json')); print('dist1', d['value'], d['ms_per_step'], {k:d[k].get('ms_per_step') for k in ('hbm_resident','steady_state','strong_scaling') if k in d})
PY
FRT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --sharded-gallery --batch 64 --gallery 1250000 --no-cpu-baseline 2> $O/bench_sharded.err | head -1 > $O/bench_sharded.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02b/bench_sharded.json')); print('sharded', d['value'], d['ms_per_step'])
PY
