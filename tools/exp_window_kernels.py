"""What the timed window of a bench command consists of, from a rocprofv3 --kernel-trace run: the window = from the start of the
n-th last env_step launch to the end of the last kernel; per kernel name its launches and time inside, per stream (queue) the
busy time, and the window's span per step.   python tools/exp_window_kernels.py <trace dir> <n> [anchor kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

d, n = sys.argv[1], int(sys.argv[2])
anchor = sys.argv[3] if len(sys.argv) > 3 else "env_step"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
ev.sort()
anchors = [e for e in ev if anchor in e[2]]
t0 = anchors[-n][0]
t1 = max(e[1] for e in ev)
inside = [e for e in ev if e[0] >= t0]
print("window: %d launches, span %.1f us = %.2f us per step" % (len(inside), (t1 - t0) / 1e3, (t1 - t0) / 1e3 / n))
by = defaultdict(lambda: [0, 0.0])
q = defaultdict(float)
for a, b, k, qu in inside:
    by[k][0] += 1
    by[k][1] += (b - a) / 1e3
    q[qu] += (b - a) / 1e3
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s %4d launches  %8.2f us each  %7.2f us per step" % (k[:44], c, t / c, t / n))
for qu, t in q.items():
    print("  queue %s busy %.2f us per step" % (qu, t / n))
# gaps on the busiest queue
main = max(q, key=q.get)
m = [e for e in inside if e[3] == main]
gaps = defaultdict(lambda: [0, 0.0])
for x, y in zip(m, m[1:]):
    gaps[x[2] + " -> " + y[2]][0] += 1
    gaps[x[2] + " -> " + y[2]][1] += (y[0] - x[1]) / 1e3
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:10]:
    print("  gap %-70s %4d x %7.2f us" % (k[:70], c, t / c))
