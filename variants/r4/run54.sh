set -uo pipefail
# the compiler's NaN canonicalisations in front of fminf / fmaxf (v_max_f32 x, x, x: half-rate instructions) taken out
# of the library's assembly: the upper bound of what min / max without them would give
for v in "" _nc "" _nc; do
  export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip$v.so
  echo -n "lib '$v' full: "; python bench.py --no-also --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python variants/benchline.py | cut -c1-260
done
export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip_nc.so
echo -n "cuvol nc: "; timeout -k 10 200 python variants/bench_cuvol.py 2>/dev/null | tr -d '\n ' | cut -c1-300; echo
timeout -k 10 600 python3 -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
