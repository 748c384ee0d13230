"""One line per bench run: value, ms/step and the per-kernel times (stdin: bench.py's JSON line)."""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["per_kernel_ms"])
