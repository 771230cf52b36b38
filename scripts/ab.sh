#!/bin/bash
# interleaved A/B of two env settings in ONE call (same device): scripts/ab.sh "ENV_A" "ENV_B" [rounds] [bench args]
A="$1"; B="$2"; R=${3:-3}; shift 3
for i in $(seq $R); do
  for cfg in "$A" "$B"; do
    env $cfg timeout -k 5 100 python bench.py --no-cpu-baseline --no-scaling-base --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'])"
  done
done
