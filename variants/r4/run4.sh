set -uo pipefail
mkdir -p gpurun_out/r4
B="python bench.py --width 72 --shuffle --no-cpu-baseline --no-verify --steps 30 --warmup 5"
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["roofline"]["per_kernel_ms"]
print(d["value"], d["ms_per_step"], {a:k[a] for a in k if a in("k_sweep","k_links_planes","k_links_checksum","k_build_masks2")})
PY
}
for rpw in 2 4; do for tm in 2 8; do for km in 4 16; do
  export ASURF_SWEEP_RPW=$rpw ASURF_LAZY_TM=$tm ASURF_LAZY_KM=$km
  $B > gpurun_out/r4/caps.json 2>gpurun_out/r4/caps.err || echo FAILED; echo -n "lazy rpw $rpw tm $tm km $km: "; line gpurun_out/r4/caps.json
done; done; done
