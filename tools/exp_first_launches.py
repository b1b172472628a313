"""The launches of an episode's first learner steps, one by one, from a rocprofv3 --kernel-trace run of bench.py: everything
between the end of reset_kernel and the n-th env-step launch behind it.   python tools/exp_first_launches.py <trace dir> [n]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
ev.sort()
i0 = max(i for i, e in enumerate(ev) if e[2].startswith("reset_kernel") and i < len(ev) - 50) if any(e[2].startswith("reset_kernel") for e in ev) else 0
# the first reset of the run
i0 = next(i for i, e in enumerate(ev) if e[2].startswith("reset_kernel"))
t0 = ev[i0][1]
steps = 0
prev_end = t0
for a, b, k, q in ev[i0 + 1:]:
    if k.startswith("env_step") or k.startswith("env_kernel"):
        steps += 1
        if steps > n:
            break
        print("---- env launch %d at +%.1f us" % (steps, (a - t0) / 1e3))
    print("  q%s %-46s %9.1f us   (gap before: %7.1f us)" % (q, k[:46], (b - a) / 1e3, (a - prev_end) / 1e3))
    prev_end = max(prev_end, b)
