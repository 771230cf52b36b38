#!/bin/bash
# Same-call A/B of library variants on the other workloads: scripts/ab_scenes.sh OUT ROUNDS v1 v2 ...  (variants/NAME.so; "cur" = librtow.so)
# Per round and variant: kernel ms of suzanne (C4), mesh100k at 256 spp (C5 shape) and the moving cover.
OUT=$1; R=$2; shift 2
for i in $(seq $R); do
  for v in "$@"; do
    lib=$PWD/raytracing-one-weekend_amd/variants/$v.so
    [ "$v" = cur ] && lib=$PWD/raytracing-one-weekend_amd/librtow.so
    line="$v"
    for sc in "suzanne" "mesh100k --spp 256" "moving"; do
      ms=$(RTOW_LIB=$lib timeout -k 10 300 python scripts/bench_scene.py $sc --steps 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['Msamples_per_s'])") || exit 1
      line="$line | ${sc%% *} $ms"
    done
    echo "$line" >> $OUT
  done
done
