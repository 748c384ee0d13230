set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_reg_gpu.py tests/test_grid_api_gpu.py -x -q -m gpu > gpurun_out/r4/t24.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t24.log
for i in 1 2; do python variants/bench_iteration.py --rays 5184 2>/dev/null | tail -1 | cut -c1-120; done
bash variants/r4/run27.sh 2>&1 | grep "tv_grad\|sum of"
