set -uo pipefail
ASURF_LIB=$PWD/alphasurf_amd/csrc/libasurf_hip_head.so timeout -k 10 200 python variants/bench_cuvol.py --no-image-width 2>/dev/null | tr -d '\n '; echo
timeout -k 10 200 python variants/bench_cuvol.py --no-image-width 2>/dev/null | tr -d '\n '; echo
timeout -k 10 200 python variants/bench_cuvol.py 2>/dev/null | tr -d '\n '; echo
