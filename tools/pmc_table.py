"""Summarise rocprofv3 --pmc results (rocpd sqlite) per kernel dispatch: python tools/pmc_table.py db [name-substring] [grid_filter]"""
import collections
import sqlite3
import sys

db, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
rows = c.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection").fetchall()
d = collections.OrderedDict()
for disp, name, grid, cn, val, dur in rows:
    if sub not in name:
        continue
    k = (disp, name.replace("(anonymous namespace)::", "").replace("void ", "")[:34], grid, dur)
    d.setdefault(k, collections.defaultdict(float))[cn] += val
names = sorted({cn for v in d.values() for cn in v})
print("disp kernel grid dur_us " + " ".join(n.replace("SQ_", "") for n in names))
for (disp, name, grid, dur), v in d.items():
    print(disp, name, grid // 256, round(dur / 1e3, 1), " ".join("%.3g" % v[n] for n in names))
