#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps && mkdir -p /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp "$(find /tmp/ps -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r02e/kernel_stats.csv')))
for r in rows[:45]: print(r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:100], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
# per-dispatch table of one recogniser pass
rm -rf /tmp/pe && mkdir -p /tmp/pe
rocprofv3 --kernel-trace -d /tmp/pe -o tr -- python $GRAFT_REPO_ROOT/tools/prof_embed.py 128 3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/trace_table.py "$(find /tmp/pe -name '*.db' | head -1)" arc_input fc_finalize | tee $O/embed_pass.txt
