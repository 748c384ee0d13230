set -uo pipefail
mkdir -p gpurun_out/r4
timeout -k 10 400 python -m pytest tests/test_cuvol_gpu.py tests/test_oracle_cuvol.py -x -q -m gpu > gpurun_out/r4/t38.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4/t38.log
timeout -k 10 200 python variants/bench_cuvol.py > gpurun_out/r4/cuvol_tiles.json 2>gpurun_out/r4/e38.err; cat gpurun_out/r4/cuvol_tiles.json
timeout -k 10 200 python variants/bench_cuvol.py --no-image-width > gpurun_out/r4/cuvol_rows.json 2>gpurun_out/r4/e38.err; cat gpurun_out/r4/cuvol_rows.json
