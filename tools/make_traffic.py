"""profiles/<tag>_pmc.csv -> profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).
HBM bytes per launch = (FETCH_SIZE x fetch_scale + WRITE_SIZE) KiB x 1024.  The guide's x2 correction for wide
streaming kernels is NOT applied: the learner kernels move 4-8 byte gathers, for which FETCH_SIZE agrees with TCC_MISS x 64 B and
with the sectors touched (profiles/r01_fetch_calibration.txt; `fetch_over_miss64` per kernel below).  The ROW-LOADING kernels --
every lane reading its own 224-byte record with 16-byte loads -- are scaled by 1.3: for that pattern FETCH_SIZE reports 0.72-0.89 x
the 64-byte sectors the lanes really touch (about four requests in ten are 128-byte requests tallied at 64:
profiles/r06_rowload_fetch_calibration.txt, tools/ubench/calibrate_rowload.sh); the uncorrected figure is kept beside it."""
import csv, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(root, "profiles", tag + "_pmc.csv")
# several instantiations of one kernel template may run (env_kernel<64;2;1> takes the step, <64;2;2> the -- usually empty --
# work list, <64;2;0> the first step): the entry of the base name is the instantiation that moves the most bytes in total
inst = {}
with open(src) as fh:
    for r in csv.DictReader(l for l in fh if not l.startswith("#")):
        d = inst.setdefault(r["kernel"], {"_launches": int(r["launches"])})
        d[r["counter"]] = float(r["avg_per_launch"])
rows = {}
for name, c in inst.items():
    k = name.split("<")[0]
    total = (c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * c["_launches"]
    if k not in rows or total > rows[k][0]:
        rows[k] = (total, name, c)
rows = {k: dict(v[2], _instantiation=v[1]) for k, v in rows.items()}
ROW_LOADERS = ("env_step_kernel", "env_step16_kernel", "env_kernel", "reset_kernel", "prepass_extend_kernel")
out = {}
for k, c in sorted(rows.items()):
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    scale = 1.3 if k in ROW_LOADERS else 1.0
    e = {"hbm_bytes_per_launch": (c["FETCH_SIZE"] * scale + c["WRITE_SIZE"]) * 1024.0, "instantiation": c["_instantiation"], "fetch_kib": c["FETCH_SIZE"],
         "write_kib": c["WRITE_SIZE"], "fetch_scale": scale, "hbm_bytes_per_launch_uncorrected": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
         "source": "profiles/%s_pmc.csv" % tag}
    for n in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
        if n in c:
            e[n.lower()] = c[n]
    # instruction issue (bench.py's issue_frac): wave-level instructions of every kind per launch, waves, and the SQ's own clocks
    insts = [c.get(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")]
    if all(v is not None for v in insts):
        e["insts_per_launch"] = sum(insts)
        e["insts_valu"], e["insts_salu"] = c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"]
    for n, key in (("SQ_WAVES", "waves"), ("SQ_WAVE_CYCLES", "wave_quad_cycles"), ("SQ_BUSY_CYCLES", "busy_cycles"), ("SQ_WAIT_ANY", "wait_any_quad_cycles"),
                   ("SQ_ACTIVE_INST_VALU", "active_valu_quad_cycles")):
        if n in c:
            e[key] = c[n]
    if c.get("TCC_MISS_sum"):
        e["fetch_over_miss64"] = round(c["FETCH_SIZE"] * 1024.0 / (c["TCC_MISS_sum"] * 64.0), 3)
    out[k] = e
out["_source"] = "profiles/%s_pmc.csv: rocprofv3 --pmc passes of the bench command (one counter group per pass), averages per launch; a static file, not measured in the bench run itself" % tag
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles", "pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
for k, e in out.items():
    if k.startswith("_"):
        continue
    print(k, round(e["hbm_bytes_per_launch"] / 1e6, 1), "MB/launch", e.get("fetch_over_miss64"))
