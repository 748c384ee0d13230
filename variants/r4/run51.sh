set -uo pipefail
OUT=gpurun_out/r4/pmc_nrm
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES \
    --output-format csv -d "$OUT/sq" -- python3 variants/bench_iteration.py --rays 5184 --iters 10 > "$OUT/sq.log" 2>&1 || echo "sq failed"
timeout -k 10 150 rocprofv3 --pmc TCC_ATOMIC_sum TCC_REQ_sum --output-format csv -d "$OUT/at" -- python3 variants/bench_iteration.py --rays 5184 --iters 10 > "$OUT/at.log" 2>&1 || echo "at failed"
timeout -k 10 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d "$OUT/s2" -- python3 variants/bench_iteration.py --rays 5184 --iters 10 > "$OUT/s2.log" 2>&1 || echo "s2 failed"
python3 - <<'PY'
import csv, glob
acc = {}
for grp in ("sq", "at", "s2"):
    for f in glob.glob(f"gpurun_out/r4/pmc_nrm/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for kn in ("k_surface_normal_grad_sparse",):
                if kn in r["Kernel_Name"]:
                    a = acc.setdefault((kn, r["Counter_Name"]), [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in sorted(acc.items()): print(k, round(v[0] / v[1], 1), flush=True)
PY
