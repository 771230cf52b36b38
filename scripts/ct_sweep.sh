#!/bin/bash
for ct in 0 1 2 3 5; do for leaf in 4 7; do
  RTOW_BVH_CT=$ct RTOW_BVH_LEAF=$leaf timeout -k 5 100 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ct $ct leaf $leaf', d['value'], d['ms_per_step'], d['config']['node_tests_per_segment'], d['config']['prim_tests_per_segment'])"
done; done
for ct in 0 2 4; do RTOW_BVH_CT=$ct RTOW_BVH_LEAF=4 timeout -k 5 100 python scripts/bench_scene.py suzanne --spp 32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('suzanne ct $ct', d['Msamples_per_s'], d['node_tests_per_segment'], d['prim_tests_per_segment'])"; done
