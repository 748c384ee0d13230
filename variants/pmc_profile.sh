#!/bin/bash
# Regenerates profiles/counters.json (what bench.py prices its per-kernel ceilings with) and the kernel-trace summary of
# the same command.  Run on the GPU box, from the repository root:
#
#     bash variants/pmc_profile.sh [tag] [extra bench.py flags ...]         # e.g.  bash variants/pmc_profile.sh r3
#
# Five rocprofv3 runs of the SAME command, the program directly after `--` (no env / bash -c / launcher in between):
#   1. --kernel-trace --stats            -> profiles/<tag>_kernel_stats.csv   (average launch duration per kernel)
#   2. --pmc FETCH_SIZE                  \  counters in their own runs, never together with a trace domain;
#   3. --pmc WRITE_SIZE                   } FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots);
#   4. --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
#   5. --pmc TCC_ATOMIC_sum               (atomic requests at the L2: one per 64-byte line an instruction touches -- what the
#                                          scatter kernels are priced against, variants/micro/atomics.hip: 21 G/s)
# plus the calibration kernels (variants/pmc_calib.py: a 512 MiB copy and fill) under passes 2 and 3, which pin the
# gfx950 unit correction (FETCH_SIZE counts 128-byte read requests at 64 B: doubled; WRITE_SIZE as is).
# variants/make_counters.py folds the four outputs into profiles/<tag>_counters.json and profiles/counters.json,
# stamped with the configuration the counters were collected on (bench.py ignores them for any other).
set -euo pipefail
TAG=${1:-r3}
shift || true
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT" profiles
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="bench.py --no-cpu-baseline --no-verify --no-also --steps 3 --warmup 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $BENCH > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$OUT/sq" -- python3 $BENCH > "$OUT/sq.log" 2>&1
rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d "$OUT/atomic" -- python3 $BENCH > "$OUT/atomic.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -- python3 variants/pmc_calib.py > "$OUT/calib_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -- python3 variants/pmc_calib.py > "$OUT/calib_write.log" 2>&1
python3 variants/make_counters.py "$OUT" "$TAG" $BENCH
# (gpurun brings back gpurun_out/ only)
mkdir -p gpurun_out/profiles && cp profiles/${TAG}_counters.json profiles/${TAG}_kernel_stats.csv profiles/counters.json gpurun_out/profiles/
