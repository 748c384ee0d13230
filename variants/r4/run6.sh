set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden_fused or aux_losses or sharding or determinism or replay_equals or fallback_kernels or record_depth or ray_ordering" > gpurun_out/r4/t6.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4/t6.log
for m in run tab; do for t in 0 1; do
  export ASURF_BWD_SH=$m ASURF_TILES=$t
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4/sh_${m}_$t.json 2>gpurun_out/r4/sh_${m}_$t.err || echo FAILED
  echo -n "$m tiles=$t full: "; python variants/benchline.py < gpurun_out/r4/sh_${m}_$t.json
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --heavy levels > gpurun_out/r4/shh_${m}_$t.json 2>gpurun_out/r4/shh_${m}_$t.err || echo FAILED
  echo -n "$m tiles=$t heavy levels: "; python variants/benchline.py < gpurun_out/r4/shh_${m}_$t.json
done; done
