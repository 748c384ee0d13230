set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden_fused or aux_losses or sharding or determinism or replay_equals or fallback_kernels or record_depth" > gpurun_out/r4/t5.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4/t5.log
for m in run tab; do
  export ASURF_BWD_SH=$m
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/sh_$m.json 2>gpurun_out/r4/sh_$m.err || echo FAILED
  echo -n "$m full: "; python variants/benchline.py < gpurun_out/r4/sh_$m.json
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --heavy levels > gpurun_out/r4/shh_$m.json 2>gpurun_out/r4/shh_$m.err || echo FAILED
  echo -n "$m heavy levels: "; python variants/benchline.py < gpurun_out/r4/shh_$m.json
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 --shuffle --width 72 > gpurun_out/r4/sh5_$m.json 2>gpurun_out/r4/sh5_$m.err || echo FAILED
  echo -n "$m 5k: "; python variants/benchline.py < gpurun_out/r4/sh5_$m.json
done
