set -uo pipefail
# every short-encoded select (v_cndmask_b32_e32 ..., vcc) of the library's device code re-assembled in the long
# encoding (_e64): what the slow re-read of VCC (variants/micro/issue_mix.hip) costs the real kernels
mkdir -p gpurun_out/r4
for v in "" _e64; do
  export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip$v.so
  echo "== lib '$v'"
  echo -n "full: "; python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
  echo -n "heavy levels: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels 2>/dev/null | python variants/benchline.py
  echo -n "heavy fake: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
  echo -n "cfg3: "; python bench.py --no-also --no-cpu-baseline --reso 512 --basis-dim 16 2>/dev/null | python variants/benchline.py
  echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
  echo -n "iteration: "; timeout -k 10 120 python3 variants/bench_iteration.py --rays 5184 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/iter', d['ms_per_iteration'])"
  echo -n "cuvol: "; timeout -k 10 200 python variants/bench_cuvol.py 2>/dev/null | tr -d '\n ' | cut -c1-400; echo
done
export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip_e64.so
timeout -k 10 600 python3 -m pytest tests/test_pipeline_gpu.py tests/test_bwd_merge_gpu.py tests/test_reg_gpu.py tests/test_cuvol_gpu.py -x -q -m gpu 2>&1 | tail -3
