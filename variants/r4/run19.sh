set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t19.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4/t19.log
echo -n "full: "; python bench.py --no-also --no-cpu-baseline 2>gpurun_out/r4/e19.err | python variants/benchline.py
echo -n "full epoch: "; python bench.py --no-also --no-cpu-baseline --links-epoch 2>gpurun_out/r4/e19.err | python variants/benchline.py
echo -n "heavy levels: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels 2>gpurun_out/r4/e19.err | python variants/benchline.py
