set -uo pipefail
python -m pytest tests/test_sum_list_gpu.py tests/test_links_epoch_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "not fuzz" 2>&1 | tail -2
echo -n "full: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 50 --warmup 10 2>/dev/null | python variants/benchline.py
echo -n "full epoch: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 50 --warmup 10 --links-epoch 2>/dev/null | python variants/benchline.py
echo -n "5k: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
echo -n "5k epoch: "; python bench.py --no-also --no-cpu-baseline --no-verify --steps 200 --warmup 20 --shuffle --width 72 --links-epoch 2>/dev/null | python variants/benchline.py
echo -n "sum_list=0 full: "; ASURF_SUM_LIST=0 python bench.py --no-also --no-cpu-baseline --no-verify --steps 50 --warmup 10 2>/dev/null | python variants/benchline.py
echo -n "cfg3: "; python bench.py --no-also --no-cpu-baseline --no-verify --reso 512 --basis-dim 16 2>/dev/null | python variants/benchline.py
