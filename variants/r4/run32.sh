set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py tests/test_bwd_merge_gpu.py -x -q -m gpu -k "not fuzz" > gpurun_out/r4/t32.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t32.log
echo -n "full: "; python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
echo -n "heavy levels: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels 2>/dev/null | python variants/benchline.py
echo -n "heavy fake: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
echo -n "cfg3: "; python bench.py --no-also --no-cpu-baseline --reso 512 --basis-dim 16 2>/dev/null | python variants/benchline.py
echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
echo -n "shuffle: "; python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 --shuffle 2>/dev/null | python variants/benchline.py
