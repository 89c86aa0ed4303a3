"""Per-dispatch table of the LAST pass in a rocprofv3 --kernel-trace db: python tools/trace_table.py db first-kernel-substring [stop-substring]"""
import sqlite3
import sys

db, first = sys.argv[1], sys.argv[2]
stop = sys.argv[3] if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if first in r[0]]
out, tot = [], 0.0
for r in rows[idx[-1]:]:
    nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if stop and stop in nm:
        break
    d = (r[2] - r[1]) / 1e3
    tot += d
    out.append("%s %.0f g(%d,%d,%d)" % (nm[-24:], d, r[3] // max(r[6], 1), r[4], r[5]))
print(" | ".join(out))
print("total us", round(tot, 1))
