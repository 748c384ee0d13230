set -uo pipefail
for n in 30 30 100 200; do echo -n "iters=$n: "; python variants/bench_iteration.py --rays 5184 --iters $n 2>/dev/null | tail -1 | cut -c1-130; done
echo -n "verify iters=30: "; python variants/bench_iteration.py --rays 5184 --iters 30 --verify 2>/dev/null | tail -1 | cut -c1-130
echo -n "verify iters=200: "; python variants/bench_iteration.py --rays 5184 --iters 200 --verify 2>/dev/null | tail -1 | cut -c1-130
