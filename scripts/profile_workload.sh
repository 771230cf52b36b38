#!/bin/bash
# Kernel-trace statistics + PMC passes for one bench workload on the GPU box, summarised into
# gpurun_out/ (copy what should be judged into profiles/).
#   scripts/profile_workload.sh TAG WORKLOAD KERNEL_USED [extra bench args]
# e.g. scripts/profile_workload.sh r02a_suzanne suzanne 2
export TMPDIR=/tmp
TAG=$1; WL=$2; KU=$3; shift 3
ARGS="--workload $WL --no-other-configs --no-end-to-end --no-reference-boundary $*"
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$TAG -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-scaling-base $ARGS > gpurun_out/stats_$TAG.log 2>&1 || { echo "stats run failed"; tail -5 gpurun_out/stats_$TAG.log; exit 1; }
cp $(ls gpurun_out/stats_$TAG/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
grep "^{\"metric\"" gpurun_out/stats_$TAG.log | tail -1 > gpurun_out/${TAG}_bench.json
bash scripts/pmc_passes.sh $TAG $ARGS || exit 1
python3 scripts/pmc_summary.py $TAG rtow_trace $WL fast $KU > gpurun_out/${TAG}_pmc.json
echo "profiled $WL -> gpurun_out/${TAG}_{kernel_stats.csv,bench.json,pmc.json}"
