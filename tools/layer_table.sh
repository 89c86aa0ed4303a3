#!/bin/bash
# per-launch durations (us) of ONE recogniser pass, in launch order, under rocprofv3 (GPU box; args: env assignments; NF = faces)
cd /tmp && export TMPDIR=/tmp
for E in "$@"; do
rm -rf /tmp/pl && mkdir -p /tmp/pl
env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -o lt -- python $GRAFT_REPO_ROOT/tools/prof_embed.py ${NF:-4} 6 > /dev/null 2>&1
echo "== $E (faces ${NF:-4})"
python - <<'PY'
import csv,glob,re
f=glob.glob('/tmp/pl/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'copyBuffer' not in r['Kernel_Name'] and 'fillBuffer' not in r['Kernel_Name']]
n=len(rows)//6
last=[rows[len(rows)-n*k-n:len(rows)-n*k] if k else rows[len(rows)-n:] for k in range(3)]
def short(s):
    s=s.replace('(anonymous namespace)::','')
    m=re.search(r'conv_small_kernelILb(\d)ELi(\d)ELi(\d)ELi(\d)',s)
    if m: return 'small<scf%s,nw%s,nt%s,d%s>'%m.groups()
    return re.sub(r'\(.*','',s)[:44]
out=[]
for i in range(n):
    d=min((int(p[i]['End_Timestamp'])-int(p[i]['Start_Timestamp']))/1e3 for p in last)
    out.append('%s %.1f'%(short(last[0][i]['Kernel_Name']),d))
print(' | '.join(out))
print('sum of minima over 3 passes: %.1f us, launches %d; span of last pass %.1f us'%(sum(float(o.split()[-1]) for o in out), n, (int(last[0][-1]['End_Timestamp'])-int(last[0][0]['Start_Timestamp']))/1e3))
PY
done
