set -uo pipefail
echo -n "heavy fake: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
echo -n "heavy levels: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels 2>/dev/null | python variants/benchline.py
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fake" 2>&1 | tail -2
