#!/bin/bash
set -uo pipefail
OUT=gpurun_out/r4/pmc_cuvol
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES \
    --output-format csv -d "$OUT/sq" -- python3 variants/bench_cuvol.py > "$OUT/sq.log" 2>&1 || echo "sq failed"
timeout -k 10 200 rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d "$OUT/at" -- python3 variants/bench_cuvol.py > "$OUT/at.log" 2>&1 || echo "at failed"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fe" -- python3 variants/bench_cuvol.py > "$OUT/fe.log" 2>&1 || echo "fe failed"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/wr" -- python3 variants/bench_cuvol.py > "$OUT/wr.log" 2>&1 || echo "wr failed"
python3 - <<'PY'
import csv, glob
acc = {}
for grp in ("sq", "at", "fe", "wr"):
    for f in glob.glob(f"gpurun_out/r4/pmc_cuvol/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for kn in ("k_cuvol_bwd", "k_cuvol_fwd"):
                if kn in r["Kernel_Name"]:
                    a = acc.setdefault((kn, r["Counter_Name"]), [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in sorted(acc.items()): print(k, round(v[0] / v[1], 1), flush=True)
PY
