set -uo pipefail
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python bench.py --width 72 --shuffle --no-cpu-baseline --no-verify --steps 30 --warmup 5"
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["roofline"]["per_kernel_ms"]
print(d["value"], d["ms_per_step"], {a:k[a] for a in k if a in("k_sweep","k_links_planes","k_links_checksum","k_build_masks2")})
PY
}
$B > gpurun_out/r4/l_split.json 2>gpurun_out/r4/l_split.err || echo FAILED; echo -n "lazy default: "; line gpurun_out/r4/l_split.json
export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip_tl.so
export ASURF_LAZY=0
for n in 1 2 3 4 5 0; do
  export ASURF_DEBUG=$((256 * n))
  $B > gpurun_out/r4/nl_ph_$n.json 2>gpurun_out/r4/ph.err || echo FAILED; echo -n "NON-lazy phases rpw 4 leave-after $n: "; line gpurun_out/r4/nl_ph_$n.json
done
unset ASURF_DEBUG ASURF_LIB ASURF_LAZY
for mode in lazy nolazy; do
  if [ $mode = nolazy ]; then export ASURF_LAZY=0; else unset ASURF_LAZY; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM \
     --output-format csv -d gpurun_out/r4/pmc_$mode -- python3 bench.py --width 72 --shuffle --no-cpu-baseline --no-verify --steps 3 --warmup 1 > gpurun_out/r4/pmc_$mode.log 2>&1 || echo PMC FAILED
done
python3 - <<'PY'
import csv, glob
for mode in ("lazy","nolazy"):
    acc = {}
    for f in glob.glob(f"gpurun_out/r4/pmc_{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sweep_wave" in r["Kernel_Name"]:
                a = acc.setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    print(mode, {k: round(v[0] / v[1] / 1296, 1) for k, v in sorted(acc.items())}, "launches", max([v[1] for v in acc.values()] or [0]), flush=True)
PY
