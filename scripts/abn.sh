#!/bin/bash
# interleaved comparison of N env settings in ONE call: scripts/abn.sh ROUNDS "ENV1" "ENV2" ...
R=$1; shift
for i in $(seq $R); do for cfg in "$@"; do
  env $cfg timeout -k 5 100 python bench.py --no-cpu-baseline --no-scaling-base --steps 8 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'])"
done; done
