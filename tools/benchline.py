"""Condense bench.py's JSON line (stdin) to one short line: tag, value, ms/step, per-kernel ms."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    k = d.get("roofline", {}).get("all_kernels_avg_ms", {})
    print(tag, round(d["value"] / 1e6, 2), d["ms_per_step"], d.get("value_amortised"), {a: b for a, b in k.items()})
