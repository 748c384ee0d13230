set -uo pipefail
for r in 0 9999999; do
  echo -n "full relaxed=$r: "; ASURF_RELAXED=$r python bench.py --no-also --no-cpu-baseline --no-verify --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
  echo -n "heavy fake relaxed=$r: "; ASURF_RELAXED=$r python bench.py --no-also --no-cpu-baseline --no-verify --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
done
