set -uo pipefail
mkdir -p gpurun_out/r4
B="python bench.py --width 72 --shuffle --no-cpu-baseline --no-verify --steps 30 --warmup 5"
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["roofline"]["per_kernel_ms"]
print(d["value"], d["ms_per_step"], {a:k[a] for a in k if a in("k_sweep","k_links_planes","k_build_masks2")})
PY
}
for rpw in 1 2 4 8; do
  export ASURF_SWEEP_RPW=$rpw
  $B > gpurun_out/r4/l_rpw$rpw.json 2>gpurun_out/r4/l_rpw$rpw.err || echo FAILED; echo -n "lazy rpw $rpw: "; line gpurun_out/r4/l_rpw$rpw.json
done
export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip_tl.so
for rpw in 4 1; do
  export ASURF_SWEEP_RPW=$rpw
  for n in 1 2 3 4 5 0; do
    export ASURF_DEBUG=$((256 * n))
    $B > gpurun_out/r4/ph_rpw${rpw}_$n.json 2>gpurun_out/r4/ph.err || echo FAILED; echo -n "phases rpw $rpw leave-after $n: "; line gpurun_out/r4/ph_rpw${rpw}_$n.json
  done
done
unset ASURF_DEBUG ASURF_LIB ASURF_SWEEP_RPW
# the headline: has the refactor of k_sweep_wave cost anything?
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/full_a.json 2>gpurun_out/r4/full_a.err || echo FAILED
python variants/benchline.py < gpurun_out/r4/full_a.json
