#!/bin/bash
# 64-channel 3x3 convs at one / four frames: fp16 hi/lo split kernel (default) against the true-fp32 MFMA kernel (FRT_DET_SPLIT=0) and 16x16 tiles (FRT_DET_SPLIT_PBW=2); tuning build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r04z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so
for B in 1 4; do for V in "A=1" "FRT_DET_SPLIT=0" "FRT_DET_SPLIT_PBW=2"; do
  rm -rf /tmp/pd && mkdir -p /tmp/pd
  env $V rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o st -- python $ROOT/tools/prof_det.py $B 6 > /dev/null 2>&1
  echo "== B=$B $V"
  grep "conv3x3" "$(find /tmp/pd -name '*kernel_stats.csv' | head -1)" | cut -c1-150
done; done > $OUT/r04z_split_ab_raw.txt 2>&1
cat $OUT/r04z_split_ab_raw.txt
