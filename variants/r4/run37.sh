set -uo pipefail
export ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip_comp4.so
echo -n "comp4+prefetch full: "; python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
echo -n "comp4+prefetch heavy levels: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy levels 2>/dev/null | python variants/benchline.py
echo -n "comp4+prefetch heavy fake: "; python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --heavy fake 2>/dev/null | python variants/benchline.py
echo -n "comp4+prefetch cfg3: "; python bench.py --no-also --no-cpu-baseline --reso 512 --basis-dim 16 2>/dev/null | python variants/benchline.py
echo -n "comp4+prefetch 5k: "; python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
