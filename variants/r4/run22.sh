set -uo pipefail
for i in 1 2 3; do
echo -n "one: "; python variants/bench_iteration.py --rays 5184 --iters 200 2>/dev/null | tail -1 | cut -c1-120
echo -n "list: "; ASURF_OPTIM=list python variants/bench_iteration.py --rays 5184 --iters 200 2>/dev/null | tail -1 | cut -c1-120
done
