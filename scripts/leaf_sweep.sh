#!/bin/bash
for leaf in 1 2 3 4 6; do
  RTOW_BVH_LEAF=$leaf timeout -k 10 120 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('leaf $leaf', d['value'], d['ms_per_step'], d['config']['node_tests_per_segment'], d['config']['prim_tests_per_segment'])"
done
