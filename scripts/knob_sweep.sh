#!/bin/bash
# one bench line per environment setting:  scripts/knob_sweep.sh WORKLOAD "ENV1=a ENV2=b" "ENV1=c" ...
# (EXTRA_ARGS="--spp 256" for the big mesh)
W=$1; shift
for e in "$@"; do
  env $e timeout -k 5 150 python bench.py --workload $W --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-end-to-end --no-reference-boundary --no-other-configs $EXTRA_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W [$e]', d['value'], d['roofline']['kernel_ms'])" || exit 1
done
