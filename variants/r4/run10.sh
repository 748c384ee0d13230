set -uo pipefail
mkdir -p gpurun_out/r4
export ASURF_BWD_SH=tab ASURF_TILES=1
python variants/timeline.py 6 --no-verify 2>gpurun_out/r4/tl.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('waves','span_us','wave_us','wave_us_mean','sum_wave_us','last_start_us','in_flight','started_by','dur_by_start_bin','cus_used','waves_per_cu','busy_us_per_cu','slowest'): print(k, d[k])
"
export ASURF_BWD_SH=run
python variants/timeline.py 6 --no-verify 2>gpurun_out/r4/tl.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('waves','span_us','wave_us','wave_us_mean','sum_wave_us','last_start_us','in_flight','started_by','dur_by_start_bin','cus_used','waves_per_cu','busy_us_per_cu'): print(k, d[k])
"
