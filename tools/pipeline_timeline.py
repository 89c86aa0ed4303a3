"""Per-queue activity of the pipelined benchmark from a rocprofv3 kernel trace (csv): busy fraction per hardware queue, stage overlap
matrix and a coarse timeline of a window in steady state.  usage: pipeline_timeline.py <kernel_trace.csv> [t0_ms] [window_ms] [bin_us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
t0_ms = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else None
win_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
bin_us = float(sys.argv[4]) if len(sys.argv) > 4 else 100.0
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ev.sort()
T0, T1 = ev[0][0], max(e[1] for e in ev)
# steady state: the middle of the trace
if t0_ms is None:  # steady state: 60 % into the strip-kernel launches
    pk = [e[0] for e in ev if "conv_patch" in e[3]]
    t0 = pk[int(len(pk) * 0.6)]
else:
    t0 = T0 + t0_ms * 1e6
t1 = t0 + win_ms * 1e6


def tag(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    for k, v in (("conv_patch", "P"), ("conv64", "c"), ("conv_s2", "s"), ("conv_glds", "g"), ("arc_input", "i"), ("fc_", "f"), ("match", "M"), ("gallery", "M"),
                 ("dwpw", "d"), ("pw_mfma", "d"), ("conv3x3", "D"), ("det_conv1", "d"), ("heads", "h"), ("decode", "h"), ("nms", "n"), ("crop", "C"),
                 ("pack", "k"), ("copyBuffer", "y"), ("to_half", "M"), ("se_", "e")):
        if k in n:
            return v
    return "?"


queues = sorted({e[2] for e in ev})
busy = defaultdict(float)
for s, e, q, n in ev:
    a, b = max(s, t0), min(e, t1)
    if b > a:
        busy[q] += b - a
print("window %.1f ms from +%.1f ms" % (win_ms, (t0 - T0) / 1e6))
for q in queues:
    print("queue %s busy %.1f %%" % (q, 100 * busy[q] / (t1 - t0)))
# union busy (any queue) and concurrency histogram
pts = []
for s, e, q, n in ev:
    a, b = max(s, t0), min(e, t1)
    if b > a:
        pts.append((a, 1))
        pts.append((b, -1))
pts.sort()
hist = defaultdict(float)
lvl, last = 0, t0
for t, d in pts:
    hist[lvl] += t - last
    last = t
    lvl += d
hist[lvl] += t1 - last
print("concurrency (kernels in flight): " + "  ".join("%d: %.1f%%" % (k, 100 * v / (t1 - t0)) for k, v in sorted(hist.items())))
nb = int((t1 - t0) / (bin_us * 1e3))
for q in queues:
    line = [" "] * nb
    for s, e, qq, n in ev:
        if qq != q or e < t0 or s > t1:
            continue
        for b in range(max(0, int((s - t0) / (bin_us * 1e3))), min(nb, int((e - t0) / (bin_us * 1e3)) + 1)):
            line[b] = tag(n)
    print("q%-3s|%s|" % (q, "".join(line)))
