set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py tests/test_bwd_merge_gpu.py -x -q -m gpu -k "not fuzz" > gpurun_out/r4/t31.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t31.log
echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
echo -n "full: "; python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
echo -n "heavy fake: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
python variants/bench_iteration.py --rays 5184 2>/dev/null | tail -1 | cut -c1-110
