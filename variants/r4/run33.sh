set -uo pipefail
for r in 0 9999999; do
  echo -n "heavy levels relaxed=$r: "; ASURF_RELAXED=$r python bench.py --no-also --no-cpu-baseline --no-verify --steps 10 --warmup 3 --heavy levels 2>/dev/null | python variants/benchline.py
  echo -n "cfg3 relaxed=$r: "; ASURF_RELAXED=$r python bench.py --no-also --no-cpu-baseline --no-verify --reso 512 --basis-dim 16 2>/dev/null | python variants/benchline.py
done
