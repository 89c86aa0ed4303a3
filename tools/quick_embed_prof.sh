#!/bin/bash
# quick per-kernel table of one recogniser run under rocprofv3 (GPU box; args: env assignments to apply, e.g. FRT_CONV_ABLATE=1)
cd /tmp && export TMPDIR=/tmp
for E in "$@"; do
rm -rf /tmp/pe && mkdir -p /tmp/pe
env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o st -- python $GRAFT_REPO_ROOT/tools/prof_embed.py ${NF:-128} 5 > /dev/null 2>&1
echo "== $E"
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pe/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/pass', round(tot/5e6,3))
for r in rows[:int(__import__('os').environ.get('NROWS','14'))]: print(' ', r['Name'].replace('(anonymous namespace)::','')[:80], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
done
