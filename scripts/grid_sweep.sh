#!/bin/bash
timeout -k 5 100 python bench.py --kernel bvh --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bvh', d['value'], d['ms_per_step'], d['config']['node_tests_per_segment'], d['config']['prim_tests_per_segment'])"
for large in 16 4; do for cpp in 1 2 4 8; do
  RTOW_GRID_LARGE=$large RTOW_GRID_CPP=$cpp timeout -k 5 100 python bench.py --kernel grid --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid large $large cpp $cpp', d['value'], d['ms_per_step'], d['config']['node_tests_per_segment'], d['config']['prim_tests_per_segment'])"
done; done
for k in bvh grid; do RTOW_GRID_LARGE=4 timeout -k 5 100 python scripts/bench_scene.py suzanne --spp 32 --kernel $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('suzanne $k', d['Msamples_per_s'], d['node_tests_per_segment'], d['prim_tests_per_segment'])"; done
