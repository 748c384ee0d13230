set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t30.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t30.log
python variants/bench_iteration.py --rays 5184 2>/dev/null | tail -1 | cut -c1-110
