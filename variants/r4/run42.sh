set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py tests/test_bwd_merge_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "not fuzz" > gpurun_out/r4/t42.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t42.log
for ct in 1 0; do
echo -n "comp_tile=$ct full: "; ASURF_COMP_TILE=$ct python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
echo -n "comp_tile=$ct heavy levels: "; ASURF_COMP_TILE=$ct python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels 2>/dev/null | python variants/benchline.py
echo -n "comp_tile=$ct heavy fake: "; ASURF_COMP_TILE=$ct python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
echo -n "comp_tile=$ct cfg3: "; ASURF_COMP_TILE=$ct python bench.py --no-also --no-cpu-baseline --reso 512 --basis-dim 16 2>/dev/null | python variants/benchline.py
done
