set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t40.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t40.log
timeout -k 10 200 python variants/bench_cuvol.py > gpurun_out/r4/r4_bench_cuvol.json 2>gpurun_out/r4/e40.err; tr -d '\n ' < gpurun_out/r4/r4_bench_cuvol.json; echo
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/cuvol_trace -- python3 variants/bench_cuvol.py > gpurun_out/r4/cuvol_trace.log 2>&1 || echo trace failed
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r4/cuvol_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open("gpurun_out/r4/r4_kernel_stats_cuvol.txt", "w") as out:
        for r in rows[:8]:
            line = f'{r["Name"][:100]} calls {r["Calls"]} avg_us {float(r["AverageNs"]) / 1e3:.1f}'
            print(line); out.write(line + "\n")
PY
