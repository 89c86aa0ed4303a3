#!/bin/bash
# quick per-launch table of one detector pass under rocprofv3 (GPU box; args: env assignments to apply one after another)
cd /tmp && export TMPDIR=/tmp
for E in "$@"; do
rm -rf /tmp/pd && mkdir -p /tmp/pd
env $E timeout 300 rocprofv3 --kernel-trace -d /tmp/pd -o t -- python $GRAFT_REPO_ROOT/tools/prof_det.py ${PB:-32} 4 > /dev/null 2>&1
echo "== $E"
python $GRAFT_REPO_ROOT/tools/trace_table.py $(find /tmp/pd -name "*.db" | head -1) det_conv1 | tr '|' '\n' | head -${NROWS:-40}
done
