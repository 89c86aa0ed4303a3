#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02h; mkdir -p $O
export FRT_LIB=$GRAFT_REPO_ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so
NF=128 NROWS=8 bash tools/quick_embed_prof.sh "X=1" "FRT_CONV_NT4_14=1" "FRT_CONV_S2=0" 2>&1 | tee $O/embed_ab.txt
