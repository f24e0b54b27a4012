"""Development aid: one-line summary of a bench.py JSON line read from stdin."""
import json
import sys

d = json.loads(sys.stdin.readlines()[-1])
r = d["roofline"]
gs = r.get("general_stage", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", "%.4e" % d["value"], "ms/step %.3f" % d["ms_per_step"],
      {k: round(v, 3) for k, v in r["per_kernel_avg_ms"].items() if v}, "general stage", {k: gs[k] for k in gs if "ms" in k or k == "frac"})
