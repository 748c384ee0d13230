set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py tests/test_bwd_merge_gpu.py -x -q -m gpu -k "not fuzz" > gpurun_out/r4/t14.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4/t14.log
python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/full14.json 2>gpurun_out/r4/full14.err || echo FAILED
echo -n "full: "; python variants/benchline.py < gpurun_out/r4/full14.json
python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels > gpurun_out/r4/hl14.json 2>gpurun_out/r4/hl14.err || echo FAILED
echo -n "heavy levels: "; python variants/benchline.py < gpurun_out/r4/hl14.json
python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 --shuffle --width 72 > gpurun_out/r4/k14.json 2>gpurun_out/r4/k14.err || echo FAILED
echo -n "5k: "; python variants/benchline.py < gpurun_out/r4/k14.json
python bench.py --no-also --no-cpu-baseline --reso 512 --basis-dim 16 > gpurun_out/r4/c314.json 2>gpurun_out/r4/c314.err || echo FAILED
echo -n "cfg3: "; python variants/benchline.py < gpurun_out/r4/c314.json
