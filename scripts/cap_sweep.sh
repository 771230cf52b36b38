#!/bin/bash
# resumable grid walk sweep: scripts/cap_sweep.sh "cap,max_open" ...   (first arg "off" = disabled)
for v in "$@"; do
  if [ "$v" = off ]; then unset RTOW_WALK_CAP; else export RTOW_WALK_CAP=$v; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-reference-boundary --no-other-configs --no-scaling-base $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cap=$v', d['value'], d['roofline']['kernel_ms'], d['config']['node_tests_per_segment'], d['config']['prim_tests_per_segment'])"
done
