#!/bin/bash
# quorum sweep of the BVH4 state machine: scripts/sm4_sweep.sh WORKLOAD "r,s,l" "r,s,l" ...
W=$1; shift
for v in "$@"; do
  RTOW_SM4_VOTES=$v timeout -k 10 200 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-reference-boundary ${SPP:+--spp $SPP} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W votes=$v', d['value'], d['roofline']['kernel_ms'], d['config']['node_tests_per_segment'], d['config']['prim_tests_per_segment'])"
done
