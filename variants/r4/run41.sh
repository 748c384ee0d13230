set -uo pipefail
for t in 4 8 16; do echo -n "tile $t: "; ASURF_CUVOL_TILE=$t timeout -k 10 200 python variants/bench_cuvol.py 2>/dev/null | tr -d '\n '; echo; done
