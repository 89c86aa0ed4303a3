#!/bin/bash
# GPU box: per-dispatch kernel trace of one recogniser run (tools/prof_embed.py F reps, env passes through): prints every dispatch of the LAST pass
# in order with its duration - the per-layer table of a path.   tools/trace_layers.sh F [reps]
F=${1:-4}; R=${2:-3}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && mkdir -p /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/tools/prof_embed.py $F $R > /tmp/tl/log 2>&1 || tail -3 /tmp/tl/log
python - "$R" <<'PY'
import csv, glob, sys
reps = int(sys.argv[1])
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]]
n = len(rows) // reps
last = rows[-n:]
tot = 0
agg = {}
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    g = "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    a = agg.setdefault((k, g), [0, 0.0])
    a[0] += 1
    a[1] += d
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
for (k, g), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s grid %-18s %3d x %8.1f us = %8.1f" % (k[:60], g, c, d / c, d))
print("dispatches %d  kernel time %.1f us  span first start -> last end %.1f us" % (len(last), tot, span))
PY
