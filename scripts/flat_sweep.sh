#!/bin/bash
# grid-shape sweep on the cover scene (static / moving): RTOW_GRID_FLAT x RTOW_GRID_CPP
for f in 1.0 1.5; do for c in 1.5 2.5 4.0 6.0; do for m in "" "--moving"; do RTOW_GRID_FLAT=$f RTOW_GRID_CPP=$c timeout -k 5 100 python bench.py --no-cpu-baseline --no-scaling-base --steps 6 --warmup 2 $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat=$f cpp=$c $m', d['value'], d['config']['node_tests_per_segment'], d['config']['prim_tests_per_segment'])"; done; done; done
