set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t30.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t30.log
python variants/bench_iteration.py --rays 5184 --verify 2>/dev/null | tail -1 | cut -c1-110
python variants/bench_iteration.py --rays 5184 2>/dev/null | tail -1 | cut -c1-110
python variants/bench_iteration.py 2>/dev/null | tail -1 | cut -c1-110
echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
echo -n "full: "; python bench.py --no-also --no-cpu-baseline 2>/dev/null | python variants/benchline.py
