#!/bin/bash
# same-call A/B of library variants on the two meshes with the walk's counters: scripts/experiments/r05_variant_scenes_ab.sh ROUNDS v1 v2 ...
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  lib=$PWD/raytracing-one-weekend_amd/variants/$v.so; [ "$v" = cur ] && lib=$PWD/raytracing-one-weekend_amd/librtow.so
  line="$v"
  for sc in "suzanne" "mesh100k --spp 256"; do
    ms=$(RTOW_LIB=$lib timeout -k 10 300 python scripts/bench_scene.py $sc --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['node_tests_per_segment'], d['prim_tests_per_segment'])") || exit 1
    line="$line | ${sc%% *} $ms"
  done
  echo "$line"
done; done
