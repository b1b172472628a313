"""Condense rocprofv3 CSV output into the two small files kept under profiles/:
<tag>_kernel_stats.csv (per-kernel calls / total / average duration) and <tag>_pmc.csv
(per kernel and counter: average value per launch).  Usage: see tools/profile.sh."""
import csv, glob, os, sys
from collections import defaultdict

tag, stats_dir, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
pmc_dirs = sys.argv[4:]


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0].replace(", ", ";")   # template arguments must not break the CSV


rows = []
for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
rows.sort(key=lambda r: -r[2])
with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats -- %s\n" % os.environ.get("PROF_CMD_STR", "python bench.py --no-cpu-baseline --no-kernel-timing").replace(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/", ""))
    fh.write("kernel,calls,total_ms,avg_us,percent\n")
    for n, c, t, a, p in rows:
        fh.write("%s,%d,%.3f,%.2f,%.2f\n" % (n, c, t / 1e6, a / 1e3, p))

acc = defaultdict(lambda: [0.0, 0])
for d in pmc_dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                acc[k][0] += float(r["Counter_Value"])
                acc[k][1] += 1
with open(os.path.join(out_dir, tag + "_pmc.csv"), "w") as fh:
    fh.write("# rocprofv3 --pmc <group> (one pass per group, no trace domains) -- same bench command; values are averages per launch\n")
    fh.write("kernel,counter,avg_per_launch,launches\n")
    for (k, c), (v, n) in sorted(acc.items()):
        fh.write("%s,%s,%.1f,%d\n" % (k, c, v / n, n))
print(open(os.path.join(out_dir, tag + "_kernel_stats.csv")).read())
print(open(os.path.join(out_dir, tag + "_pmc.csv")).read())
