set -uo pipefail
python -m pytest tests/test_hip_parity.py tests/test_pipeline_gpu.py -x -q -m gpu -k "not fuzz" 2>&1 | tail -2
for i in 1 2; do echo -n "full: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 50 --warmup 10 2>/dev/null | python variants/benchline.py; done
echo -n "cfg3: "; python bench.py --no-also --no-cpu-baseline --no-verify --reso 512 --basis-dim 16 2>/dev/null | python variants/benchline.py
echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
