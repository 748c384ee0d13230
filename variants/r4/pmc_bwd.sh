#!/bin/bash
# PMC counters of k_bwd_sh / k_bwd_geo: the default (image tiles + chunks sorted by voxel) and round 3's form
set -uo pipefail
OUT=gpurun_out/r4/pmc_bwd
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for form in new old; do
    if [ $form = old ]; then export ASURF_BWD_SORT=0 ASURF_BWD_TILE=0; fi
    timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES \
        --output-format csv -d "$OUT/sq_$form" -- python3 bench.py --no-cpu-baseline --no-verify --no-also --steps 3 --warmup 1 "$@" > "$OUT/sq_$form.log" 2>&1 || echo "sq $form failed"
    timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/wr_$form" -- python3 bench.py --no-cpu-baseline --no-verify --no-also --steps 3 --warmup 1 "$@" > "$OUT/wr_$form.log" 2>&1 || echo "wr $form failed"
    timeout -k 10 150 rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d "$OUT/at_$form" -- python3 bench.py --no-cpu-baseline --no-verify --no-also --steps 3 --warmup 1 "$@" > "$OUT/at_$form.log" 2>&1 || echo "at $form failed"
    echo "$form done"
done
python3 - <<'PY'
import csv, glob
for form in ("new", "old"):
    acc = {}
    for grp in ("sq", "wr", "at"):
        for f in glob.glob(f"gpurun_out/r4/pmc_bwd/{grp}_{form}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                for kn in ("k_bwd_sh", "k_bwd_geo"):
                    if kn in r["Kernel_Name"]:
                        a = acc.setdefault((kn, r["Counter_Name"]), [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(acc.items()): print(form, k, round(v[0] / v[1], 1), flush=True)
PY
