"""Registers / scratch / LDS / occupancy of every kernel of the engine library, as the compiler reports
them (-Rpass-analysis=kernel-resource-usage; device pass only, nothing is written next to the sources).
    python tools/kernel_resources.py [--csv profiles/r03_resources.csv] [extra hipcc flags]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rl_markets_amd", "csrc")
csv_path = None
if "--csv" in sys.argv:
    i = sys.argv.index("--csv")
    csv_path = sys.argv[i + 1]
    del sys.argv[i:i + 2]
# (the library's four device translation units, rl_markets_amd/csrc/lob_launch.h, compiled side by side)
units = ["lob_engine.hip", "lob_tu_env.hip", "lob_tu_prepass.hip", "lob_tu_learn.hip"]
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-c",
                           "-Rpass-analysis=kernel-resource-usage", "-Wno-unused-value", "-o", "/dev/null", u] + sys.argv[1:],
                          cwd=CSRC, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for u in units]
out = "".join(pr.communicate()[1] for pr in procs)
try:
    filt = subprocess.run(["c++filt"], input=out, stdout=subprocess.PIPE, text=True).stdout
except FileNotFoundError:
    filt = out
cur = {}
rows = []
for line in filt.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?)(?: \[-Rpass)", line) or re.search(r"remark: +(.*?)(?: \[-Rpass)", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            name = re.sub(r"\(.*", "", cur["name"])
            print("%-58s vgpr %4s agpr %3s sgpr %4s (spilled %4s) scratch %5s occ %2s lds %6s" % (
                name[:58], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("SGPRs Spill"), cur.get("ScratchSize [bytes/lane]"),
                cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
            rows.append((name.replace("void ", "").replace(", ", ";"), cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("SGPRs Spill"),
                         cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
if csv_path:
    with open(csv_path, "w") as fh:
        fh.write("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage of the four translation units of rl_markets_amd/csrc (tools/kernel_resources.py);"
                 " occupancy = waves per SIMD the register / LDS budget allows; dynamic LDS (the fast learner kernels) is not included\n")
        fh.write("kernel,vgprs,agprs,sgprs,sgprs_spilled,vgprs_spilled,scratch_bytes_per_lane,occupancy_waves_per_simd,static_lds_bytes_per_block\n")
        for r in rows:
            fh.write(",".join(str(x) for x in r) + "\n")
