set -uo pipefail
mkdir -p gpurun_out/r4
python variants/timeline.py 6 --no-verify > gpurun_out/r4/tl6.json 2>gpurun_out/r4/tl6.err; tail -1 gpurun_out/r4/tl6.json
python variants/timeline.py 7 --no-verify > gpurun_out/r4/tl7.json 2>gpurun_out/r4/tl7.err; tail -1 gpurun_out/r4/tl7.json
