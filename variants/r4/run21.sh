set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu -k "optim or mirror or train or iteration or sparse_grid or reg" > gpurun_out/r4/t21.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4/t21.log
python variants/bench_iteration.py --rays 5184 --iters 100 --verify 2>gpurun_out/r4/e21.err | tail -1
ASURF_OPTIM=list python variants/bench_iteration.py --rays 5184 --iters 100 2>gpurun_out/r4/e21.err | tail -1
python variants/bench_iteration.py --rays 5184 --iters 100 --graph 2>gpurun_out/r4/e21.err | tail -1
python variants/bench_iteration.py --iters 30 2>gpurun_out/r4/e21.err | tail -1
ASURF_OPTIM=list python variants/bench_iteration.py --iters 30 2>gpurun_out/r4/e21.err | tail -1
