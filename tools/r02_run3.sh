#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
export FRT_LIB=$GRAFT_REPO_ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so
for NF in 4 16 32 64; do
  echo "##### F=$NF"
  NF=$NF NROWS=4 bash tools/quick_embed_prof.sh "FRT_CONV_SMALL_BATCH=0" "FRT_CONV_SMALL_BATCH=1" 2>&1 | grep -E "==|total|conv_patch"
done > $O/small_batch.log 2>&1
cat $O/small_batch.log
