set -uo pipefail
# the LDS table of the regularisers (RegAcc): its size against the normal loss's atomics and time
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "" _s4096 _s1024; do
  export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip$v.so
  echo "== lib '$v'"
  timeout -k 10 120 python3 variants/bench_iteration.py --rays 5184 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/iter', d['ms_per_iteration'])"
  rm -rf gpurun_out/r4/reg$v
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/reg$v/kt -- python3 variants/bench_iteration.py --rays 5184 --iters 100 > gpurun_out/r4/reg$v.log 2>&1 || echo "trace failed"
  timeout -k 10 150 rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d gpurun_out/r4/reg$v/at -- python3 variants/bench_iteration.py --rays 5184 --iters 10 >> gpurun_out/r4/reg$v.log 2>&1 || echo "pmc failed"
  V=$v python3 - <<'PY'
import csv, glob, os
v = os.environ["V"]
for f in glob.glob(f"gpurun_out/r4/reg{v}/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_surface_normal", "k_tv_grad_sparse")):
            print("  us", round(float(r["AverageNs"]) / 1e3, 1), r["Name"][:50])
acc = {}
for f in glob.glob(f"gpurun_out/r4/reg{v}/at/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for kn in ("k_surface_normal_grad_sparse", "k_tv_grad_sparse<false>", "k_tv_grad_sparse<true>"):
            if kn in r["Kernel_Name"]:
                a = acc.setdefault(kn, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for k, a in acc.items(): print("  atomic requests", k, round(a[0] / a[1]))
PY
done
