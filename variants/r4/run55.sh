set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t55_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t55_tests.log
for i in 1 2; do echo -n "full: "; python bench.py --no-also --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python variants/benchline.py | cut -c1-330; done
echo -n "shuffle: "; python bench.py --no-also --no-cpu-baseline --shuffle 2>/dev/null | python variants/benchline.py | cut -c1-330
echo -n "iteration view: "; timeout -k 10 120 python3 variants/bench_iteration.py 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/iter', d['ms_per_iteration'])"
