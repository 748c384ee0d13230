set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t12.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r4/t12.log
( time python bench.py > gpurun_out/r4/default12.json 2>gpurun_out/r4/default12.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4/default12.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verify"]["ok"], d["roofline"]["per_kernel_ms"])
for k,v in (d.get("also") or {}).items():
    print(k, {a:v[a] for a in v if a in ("ok","value","ms_per_step","ms_per_iteration","child_s","rc","error","stderr_tail")})
PY
python bench.py --reso 512 --basis-dim 16 --eikonal --no-cpu-baseline > gpurun_out/r4/cfg3_eik.json 2>gpurun_out/r4/cfg3_eik.err || echo FAILED
echo -n "cfg3+eikonal: "; python variants/benchline.py < gpurun_out/r4/cfg3_eik.json
python bench.py --reso 512 --basis-dim 16 --no-cpu-baseline > gpurun_out/r4/cfg3.json 2>gpurun_out/r4/cfg3.err || echo FAILED
echo -n "cfg3: "; python variants/benchline.py < gpurun_out/r4/cfg3.json
