#!/bin/bash
# usage: detprof.sh TAG B [ENV=VAL ...]
R=$GRAFT_REPO_ROOT; TAG=$1; B=$2; shift 2
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pd; mkdir -p /tmp/pd
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o st -- python $R/tools/prof_det.py $B 20 > /dev/null 2>&1
cp "$(find /tmp/pd -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/det_$TAG.csv
