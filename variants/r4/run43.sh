set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t43.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t43.log
echo -n "full: "; python bench.py --no-also --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python variants/benchline.py
echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
echo -n "heavy fake: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
python variants/bench_iteration.py --rays 5184 2>/dev/null | tail -1 | cut -c1-110
python variants/bench_iteration.py --rays 5184 2>/dev/null | tail -1 | cut -c1-110
