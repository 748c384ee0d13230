#!/bin/bash
# Dynamic instruction counts of k_sweep's phases: the instrumented build (make -C alphasurf_amd/csrc timeline) with
# ASURF_DEBUG = 256 * n makes every wave leave after phase n - 1 (results invalid); one --pmc pass per n, the differences
# are the phases.  Run on the GPU box from the repository root; writes gpurun_out/sweep_phases/.
set -euo pipefail
OUT=gpurun_out/sweep_phases
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip_tl.so
for n in 0 1 2 3 4 5; do
    export ASURF_DEBUG=$((256 * n))
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM \
        --output-format csv -d "$OUT/n$n" -- python3 bench.py --no-cpu-baseline --no-verify --no-also --steps 3 --warmup 1 > "$OUT/n$n.log" 2>&1
done
python3 - <<'PY'
import csv, glob
for n in range(6):
    acc = {}
    for f in glob.glob(f"gpurun_out/sweep_phases/n{n}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_sweep_wave" in r["Kernel_Name"]:
                a = acc.setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    print(n, {k: round(v[0] / v[1] / 40000, 1) for k, v in sorted(acc.items())}, flush=True)
PY
