#!/bin/bash
# sweep of max_open (lanes still walking at which a resumable walk may stop), release rule off / on
W=$1; shift
for cfg in "0 4,16" "0 4,20" "0 4,24" "0 4,28" "0 4,32" "0 4,40" "5 4,64" "4 4,64" "6 4,64"; do
  set -- $cfg
  k=$1; cap=$2
  env RTOW_WALK_RELEASE=$k RTOW_WALK_CAP=$cap timeout -k 5 150 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-configs $EXTRA_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W release=$k cap=$cap', d['value'], d['roofline']['kernel_ms'])" || exit 1
done
