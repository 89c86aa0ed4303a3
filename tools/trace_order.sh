#!/bin/bash
# GPU box: every dispatch of the LAST pass of tools/prof_embed.py F reps, in launch order, with its duration.   tools/trace_order.sh F [reps]
F=${1:-128}; R=${2:-3}
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
for i, r in enumerate(rows[-n:]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%2d %-62s grid %-8s %7.1f us" % (i, k[:62], r.get("Grid_Size_X", r.get("Grid_Size", "?")), d))
PY
