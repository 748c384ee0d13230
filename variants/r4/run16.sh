set -uo pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/t16.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4/t16.log
bash variants/r4/pmc_bwd.sh
