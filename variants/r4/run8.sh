set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden_fused or aux_losses or sharding or determinism or replay_equals or fallback_kernels or record_depth or ray_ordering" > gpurun_out/r4/t8.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4/t8.log
for t in 0 1; do
  export ASURF_BWD_SH=tab ASURF_TILES=$t
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/sh3_$t.json 2>gpurun_out/r4/sh3_$t.err || echo FAILED
  echo -n "tab tiles=$t full: "; python variants/benchline.py < gpurun_out/r4/sh3_$t.json
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --heavy levels > gpurun_out/r4/shh3_$t.json 2>gpurun_out/r4/shh3_$t.err || echo FAILED
  echo -n "tab tiles=$t heavy levels: "; python variants/benchline.py < gpurun_out/r4/shh3_$t.json
done
python bench.py --no-cpu-baseline --steps 30 --warmup 5 --shuffle --width 72 > gpurun_out/r4/sh35.json 2>gpurun_out/r4/sh35.err || echo FAILED
echo -n "tab 5k: "; python variants/benchline.py < gpurun_out/r4/sh35.json
