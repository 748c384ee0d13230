"""k_sweep launch-shape sweep (measurement)."""
import os, sys, subprocess, json
for thr in ("128", "256", "512"):
    for noclds in ("256", "512", "1024"):
        env = dict(os.environ, ASURF_SWEEP_THREADS=thr, ASURF_SWEEP_LANES=noclds)
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "10"] + sys.argv[1:], env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(thr, noclds, j["value"], j["roofline"]["per_kernel_ms"]["k_sweep"], j["roofline"]["per_kernel_ms"]["k_geom"], flush=True)
        except Exception as e:
            print(thr, noclds, "failed", out.stderr[-300:])
