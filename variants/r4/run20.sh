set -uo pipefail
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for cfg in "5k:--rays 5184" "640k:"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rm -rf gpurun_out/r4/it$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/it$tag -- python3 variants/bench_iteration.py $fl --iters 50 > gpurun_out/r4/it$tag.log 2>&1 || echo failed
  echo "== $tag"
  python3 - $tag <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for f in glob.glob(f"gpurun_out/r4/it{tag}/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
done
