"""Run bench.py with the given extra args and print value + roofline per class in one line (helper for A/B runs)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True).stdout.strip().splitlines()
d = json.loads(out[-1])
r = d.get("roofline") or {}
print("%.1f img/s, roofline %.3f, %s" % (d["value"], r.get("frac", float("nan")), {k: round(v["avg_us"], 1) for k, v in (r.get("by_kernel") or {}).items()}))
