set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_bwd_merge_gpu.py tests/test_dist_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "not fuzz" > gpurun_out/r4/t11.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4/t11.log
python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/full11.json 2>gpurun_out/r4/full11.err || echo FAILED
echo -n "full: "; python variants/benchline.py < gpurun_out/r4/full11.json
python bench.py --no-cpu-baseline --steps 30 --warmup 5 --shuffle > gpurun_out/r4/shuf11.json 2>gpurun_out/r4/shuf11.err || echo FAILED
echo -n "shuffle: "; python variants/benchline.py < gpurun_out/r4/shuf11.json
for x in pipelined auto dense; do
ASURF_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --width 400 --steps 3 --warmup 1 --no-cpu-baseline --exchange $x > gpurun_out/r4/gloo2_$x.json 2> gpurun_out/r4/gloo2_$x.err; echo "gloo2 $x rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4/gloo2_$x.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("exchange"))
PY
done
