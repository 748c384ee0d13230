set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py tests/test_pipeline_gpu.py -x -q -m gpu -k "not fuzz" > gpurun_out/r4/t28.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t28.log
for w in 72 200 320 400 512; do
for r in 0 9999999; do
  echo -n "width=$w relaxed=$r: "; ASURF_RELAXED=$r python bench.py --no-also --no-cpu-baseline --no-verify --steps 100 --warmup 10 --shuffle --width $w 2>/dev/null | python variants/benchline.py
done; done
