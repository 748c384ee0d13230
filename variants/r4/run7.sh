set -uo pipefail
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for m in run tab; do
  export ASURF_BWD_SH=$m ASURF_TILES=1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES \
     --output-format csv -d gpurun_out/r4/pmc_sh_$m -- python3 bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 > gpurun_out/r4/pmc_sh_$m.log 2>&1 || echo PMC FAILED
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_FLAT SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_VALU \
     --output-format csv -d gpurun_out/r4/pmc2_sh_$m -- python3 bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 > gpurun_out/r4/pmc2_sh_$m.log 2>&1 || echo PMC2 FAILED
done
python3 - <<'PY'
import csv, glob
for mode in ("run","tab"):
  for pre in ("pmc","pmc2"):
    acc = {}
    for f in glob.glob(f"gpurun_out/r4/{pre}_sh_{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_bwd_sh" in r["Kernel_Name"]:
                a = acc.setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    print(mode, pre, {k: round(v[0] / v[1]) for k, v in sorted(acc.items())}, flush=True)
PY
