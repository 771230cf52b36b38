#!/bin/bash
# sweep of the walk-suspension release rule: RTOW_WALK_RELEASE=k (a walk stops only once released*4 >= held*k),
# RTOW_WALK_CAP=cap,open.   usage: [EXTRA_ARGS="--spp 256"] scripts/release_sweep.sh WORKLOAD
W=$1; shift
for cfg in "0 -" "4 -" "4 4,64" "2 4,64" "8 4,64" "4 3,64" "4 6,64" "4 2,64" "6 4,64" "3 4,64" "4 8,64"; do
  set -- $cfg
  k=$1; cap=$2
  if [ "$cap" = "-" ]; then capenv="RTOW_X=1"; else capenv="RTOW_WALK_CAP=$cap"; fi
  env RTOW_WALK_RELEASE=$k $capenv timeout -k 5 150 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-configs $EXTRA_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W release=$k cap=$cap', d['value'], d['roofline']['kernel_ms'])" || exit 1
done
