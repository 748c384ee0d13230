#!/bin/bash
# Round-4 profile set (round 3: the same with TAG r3) (run on the GPU box from the repository root): PMC counters + kernel trace of the headline
# command first (bench.py then prices its ceilings with them), then one bench line per configuration.
set -uo pipefail
TAG=${1:-r4}
O=gpurun_out/profiles
mkdir -p $O
bash variants/pmc_profile.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1 || tail -5 gpurun_out/pmc_$TAG.log
bash variants/pmc_profile.sh ${TAG}cfg3 --reso 512 --basis-dim 16 > gpurun_out/pmc_${TAG}cfg3.log 2>&1 || tail -5 gpurun_out/pmc_${TAG}cfg3.log
bash variants/pmc_profile.sh ${TAG}cfg5 --reso 768 --basis-dim 16 --band 16 --width 1600 > gpurun_out/pmc_${TAG}cfg5.log 2>&1 || tail -5 gpurun_out/pmc_${TAG}cfg5.log
cp profiles/counters.json profiles/${TAG}cfg3_counters.json profiles/${TAG}cfg5_counters.json profiles/${TAG}cfg3_kernel_stats.csv profiles/${TAG}cfg5_kernel_stats.csv gpurun_out/profiles/ 2>/dev/null || true
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-also "$@" > $O/${TAG}_bench_$name.json 2> gpurun_out/bench_$name.err || echo "FAILED $name"; python variants/benchline.py < $O/${TAG}_bench_$name.json 2>/dev/null || true; }
# the driver's command: the headline with its `also` block (5 k rays, training iteration, heavy scenes: child runs)
timeout -k 10 400 python bench.py > $O/${TAG}_bench_default_with_also.json 2> gpurun_out/bench_default.err || echo "FAILED default"
run full
run full_links_epoch --links-epoch
run shuffle --shuffle
run shuffle_as_given --shuffle --order-rays 1
run 5k --width 72 --shuffle
run 5k_links_epoch --width 72 --shuffle --links-epoch
run cfg3 --reso 512 --basis-dim 16
run cfg3_eikonal --reso 512 --basis-dim 16 --eikonal --no-cpu-baseline
run cfg3_links_epoch --reso 512 --basis-dim 16 --links-epoch
run cfg5 --reso 768 --basis-dim 16 --band 16 --width 1600 --cpu-rays 20000
run cfg5_links_epoch --reso 768 --basis-dim 16 --band 16 --width 1600 --links-epoch --no-cpu-baseline
run heavy_levels --heavy levels
run heavy_fake --heavy fake --no-cpu-baseline
for a in "--rays 5184 --verify" "--rays 5184 --graph" "--rays 5184 --graph --graph-static" "--rays 100000" "" "--graph"; do timeout -k 10 200 python variants/bench_iteration.py $a 2>/dev/null | tail -1; done > $O/${TAG}_bench_training_iteration.json
python variants/host_overhead.py 2>/dev/null | tail -3 > $O/${TAG}_host_overhead.txt
./variants/micro/issue > $O/${TAG}_micro_issue.txt 2>&1
./variants/micro/issue_mix > $O/${TAG}_micro_issue_mix.txt 2>&1 || true
./variants/micro/atomics > $O/${TAG}_micro_atomics.txt 2>&1 || true
timeout -k 10 300 python variants/bench_eval.py > $O/${TAG}_bench_eval_renders.json 2>/dev/null || true
cat $O/${TAG}_bench_training_iteration.json $O/${TAG}_host_overhead.txt
