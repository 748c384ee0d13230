set -uo pipefail
echo -n "full: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
