set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_cuvol_gpu.py tests/test_oracle_cuvol.py tests/test_grid_api_gpu.py -x -q -m gpu > gpurun_out/r4/t45.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4/t45.log
for w in 72 141 283 800; do echo -n "width $w: "; timeout -k 10 200 python variants/bench_cuvol.py --width $w 2>/dev/null | tr -d '\n '; echo; done
timeout -k 10 200 python variants/bench_cuvol.py > gpurun_out/r4/r4_bench_cuvol.json 2>/dev/null
