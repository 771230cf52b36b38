#!/bin/bash
# occupancy / block-size sweep for the BVH kernel (experiment knobs are env vars)
for blk in 256 512 1024; do for bpc in 1 2 3 4 6; do
  RTOW_BVH_BLOCK=$blk RTOW_BLOCKS_PER_CU=$bpc timeout -k 10 120 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('block $blk bpc $bpc', d['value'], d['ms_per_step'])"
done; done
