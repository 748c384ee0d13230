set -uo pipefail
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/r4/it5k_b
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/it5k_b -- python3 variants/bench_iteration.py --rays 5184 --iters 200 > gpurun_out/r4/it5k_b.log 2>&1 || echo failed
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r4/it5k_b/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_build_masks2' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp']); prev = t0; tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:8.1f} +{(e-s)/1e3:6.1f} gap {(s-prev)/1e3:5.1f}  {r['Kernel_Name'][:80]}")
    prev = e; tot += e - s
print("sum of kernels", tot / 1e3, "span", (int(rows[b]['Start_Timestamp']) - t0) / 1e3)
PY
