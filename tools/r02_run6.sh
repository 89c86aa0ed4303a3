#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embedder.py tests/test_gpu_match.py tests/test_gpu_detector.py -x -q -m gpu > $O/pytest.log 2>&1; tail -8 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe && mkdir -p /tmp/pe
rocprofv3 --kernel-trace -d /tmp/pe -o tr -- python $GRAFT_REPO_ROOT/tools/prof_embed.py 128 3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/trace_table.py "$(find /tmp/pe -name '*.db' | head -1)" arc_input fc_finalize | tr '|' '\n' | grep -E "s2|glds|total" | tee $O/embed_pass.txt
export FRT_LIB=$GRAFT_REPO_ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so
NROWS=6 bash $GRAFT_REPO_ROOT/tools/quick_det_prof.sh "FRT_ROW4_NS=2" "FRT_ROW4_NS=3" "FRT_ROW4_NS=4" 2>&1 | tee $O/det_ns.txt
unset FRT_LIB
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 50 --no-cpu-baseline --stage-profile $O/stages.json 2> $O/bench.err | head -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f/bench.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'resident', d['hbm_resident']['ms_per_step'], 'roofline', r['frac'], r['avg_launch_us'])
print({k:v['avg_launch_us'] for k,v in r['all_3x3_conv_kernels']['per_kernel'].items()})
s=json.load(open('gpurun_out/r02f/stages.json')); print({k:round(v['ms_per_step'],3) for k,v in s.items()})
PY
