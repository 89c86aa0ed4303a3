#!/bin/bash
# GPU box (env TARGET='prof_det.py 32 2' / KFILTER=regex to change the run and the kernels shown): one rocprofv3 --pmc pass per argument (a comma-free, space-separated counter group in quotes) over the recogniser run,
# printing per-kernel medians for the conv kernels.  No trace domains are combined with --pmc (pool rule).
cd /tmp && export TMPDIR=/tmp
for G in "$@"; do
rm -rf /tmp/pm && mkdir -p /tmp/pm
timeout 300 rocprofv3 --pmc $G -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/${TARGET:-prof_embed.py 128 2} > /tmp/pm/log 2>&1 || tail -3 /tmp/pm/log
python - <<'PY'
import collections, glob, sqlite3, statistics
dbs = glob.glob("/tmp/pm/**/*.db", recursive=True)
if not dbs:
    print("no db"); raise SystemExit
c = sqlite3.connect(dbs[0])
d = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for disp, name, cn, val in c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
    k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d[k][cn][disp] += val
for k in d:
    import os, re
    if re.search(os.environ.get("KFILTER", "conv_patch_kernel<2, 5, 5, false, 0, false, 7|conv64_kernel<0"), k):
        print(k, {cn: round(statistics.median(v.values())) for cn, v in d[k].items()})
PY
done
