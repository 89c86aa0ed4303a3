#!/bin/bash
# A/B: conv_dw MFMA kernels with their chunks' loads in flight (FRT_DWPW_PRE) at larger batches; tuning build, detector-only kernel traces
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r04z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FRT_LIB=$ROOT/face-recognition-cpp-tensorrt_amd/libfrt_tuning.so
for B in 32 8 4; do for PRE in 1 2; do
  rm -rf /tmp/pd && mkdir -p /tmp/pd
  FRT_DWPW_PRE=$PRE rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o st -- python $ROOT/tools/prof_det.py $B 6 > /dev/null 2>&1
  echo "== B=$B FRT_DWPW_PRE=$PRE"
  grep dwpw_mfma_kernel "$(find /tmp/pd -name '*kernel_stats.csv' | head -1)" | awk -F'","' '{printf "%s calls %s avg %.1f us\n", $1, $2, $4/1000}' | sed 's/(anonymous namespace):://g' | cut -c1-140
done; done > $OUT/r04z_dwpw_pre_ab.txt 2>&1
cat $OUT/r04z_dwpw_pre_ab.txt
