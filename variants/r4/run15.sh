set -uo pipefail
mkdir -p gpurun_out/r4
for cfg in "1:8" "1:16" "1:4" "1:32" "0:0"; do
  srt=${cfg%%:*}; t=${cfg##*:}
  echo -n "sort=$srt tile=$t full: "; ASURF_BWD_SORT=$srt ASURF_BWD_TILE=$t python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r4/e15.err | python variants/benchline.py
done
for cfg in "1:8" "1:16"; do
  srt=${cfg%%:*}; t=${cfg##*:}
  echo -n "sort=$srt tile=$t heavy levels: "; ASURF_BWD_SORT=$srt ASURF_BWD_TILE=$t python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels 2>gpurun_out/r4/e15.err | python variants/benchline.py
  echo -n "sort=$srt tile=$t heavy fake: "; ASURF_BWD_SORT=$srt ASURF_BWD_TILE=$t python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>gpurun_out/r4/e15.err | python variants/benchline.py
  echo -n "sort=$srt tile=$t cfg3: "; ASURF_BWD_SORT=$srt ASURF_BWD_TILE=$t python bench.py --no-also --no-cpu-baseline --reso 512 --basis-dim 16 2>gpurun_out/r4/e15.err | python variants/benchline.py
done
