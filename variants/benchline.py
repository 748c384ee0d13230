"""One line per bench run (stdin: bench.py's JSON line): value, ms/step, dominant kernel and its ceiling, per-kernel ms."""
import json, sys
d = json.load(sys.stdin)
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r.get("bound"), r.get("frac"), (d.get("verify") or {}).get("ok"),
      r["per_kernel_ms"])
