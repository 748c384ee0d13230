set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t_final.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4/t_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>gpurun_out/r4/bench_final.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], {k:(v.get('value') or v.get('ms_per_iteration')) for k,v in d['also'].items()}, d['verify']['ok'])"
