"""k_sweep timing probes: ASURF_DEBUG bits 32/64/128/256 end the kernel after phase 0/1/2/3a (results invalid)."""
import os, sys, subprocess, json
for dbg in (32, 64, 128, 256, 0):
    env = dict(os.environ, ASURF_DEBUG=str(dbg))
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "10"] + sys.argv[1:], env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(dbg, j["roofline"]["per_kernel_ms"], flush=True)
    except Exception as e:
        print(dbg, "failed", out.stderr[-500:])
