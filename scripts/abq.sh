#!/bin/bash
# quick interleaved comparison of N env settings in ONE call (headline workload only, no extras):
#   scripts/abq.sh ROUNDS "ENV1" "ENV2" ...      (BENCH_ARGS="--spp 500" for other bench arguments)
R=$1; shift
for i in $(seq $R); do for cfg in "$@"; do
  env $cfg timeout -k 5 100 python bench.py --no-cpu-baseline --no-scaling-base --no-other-configs --no-end-to-end --no-reference-boundary --steps 8 --warmup 2 $BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
