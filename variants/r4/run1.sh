set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_links_planes_gpu.py tests/test_links_epoch_gpu.py -x -q -m gpu > gpurun_out/r4/t1.log 2>&1; echo "tests1 rc=$?"; tail -15 gpurun_out/r4/t1.log
for lz in 0 default; do
  for ep in "" "--links-epoch"; do
    if [ $lz = default ]; then unset ASURF_LAZY; else export ASURF_LAZY=$lz; fi
    timeout -k 10 200 python bench.py --width 72 --shuffle $ep --no-cpu-baseline > gpurun_out/r4/b5k_${lz}_${ep#--}.json 2> gpurun_out/r4/b5k_${lz}_${ep#--}.err || echo FAILED
    python variants/benchline.py < gpurun_out/r4/b5k_${lz}_${ep#--}.json
  done
done
