set -uo pipefail
mkdir -p gpurun_out/r4
for x in pipelined auto dense; do
ASURF_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --width 400 --steps 3 --warmup 1 --no-cpu-baseline --exchange $x > gpurun_out/r4/gloo2_$x.json 2> gpurun_out/r4/gloo2_$x.err; echo "gloo2 $x rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4/gloo2_$x.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("exchange"))
PY
done
