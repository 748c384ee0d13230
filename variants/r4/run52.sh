set -uo pipefail
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python3 -m pytest tests/test_reg_gpu.py -x -q -m gpu 2>&1 | tail -3


for v in ${LIBS:-""}; do
  [ "$v" = "-" ] && v=""
  export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip$v.so
  echo "== lib '$v'"
  timeout -k 10 120 python3 variants/bench_iteration.py --rays 5184 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/iter', d['ms_per_iteration'])"
  rm -rf gpurun_out/r4/nrm$v
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/nrm$v/kt -- python3 variants/bench_iteration.py --rays 5184 --iters 100 > gpurun_out/r4/nrm$v.log 2>&1 || echo "trace failed"
  V=$v python3 - <<'PY'
import csv, glob, os
v = os.environ["V"]
for f in glob.glob(f"gpurun_out/r4/nrm{v}/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_surface_normal",)):
            print("  us", round(float(r["AverageNs"]) / 1e3, 1), r["Name"][:50])
PY
done
