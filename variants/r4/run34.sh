set -uo pipefail
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_sum_list_gpu.py tests/test_links_epoch_gpu.py tests/test_links_planes_gpu.py -x -q -m gpu > gpurun_out/r4/t34.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4/t34.log
for rd in "" 0; do
  echo -n "run_desc=$rd full: "; ASURF_RUN_DESC=$rd python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python variants/benchline.py
  echo -n "run_desc=$rd full epoch: "; ASURF_RUN_DESC=$rd python bench.py --no-also --no-cpu-baseline --steps 30 --warmup 5 --links-epoch 2>/dev/null | python variants/benchline.py
  echo -n "run_desc=$rd 5k: "; ASURF_RUN_DESC=$rd python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 2>/dev/null | python variants/benchline.py
  echo -n "run_desc=$rd 5k epoch: "; ASURF_RUN_DESC=$rd python bench.py --no-also --no-cpu-baseline --steps 200 --warmup 20 --shuffle --width 72 --links-epoch 2>/dev/null | python variants/benchline.py
done
