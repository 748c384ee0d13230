set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t23.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4/t23.log
for i in 1 2; do python variants/bench_iteration.py --rays 5184 --iters 200 2>/dev/null | tail -1 | cut -c1-120; done
python variants/bench_iteration.py --rays 5184 --iters 200 --graph 2>/dev/null | tail -1 | cut -c1-120
python variants/bench_iteration.py --iters 50 2>/dev/null | tail -1 | cut -c1-120
