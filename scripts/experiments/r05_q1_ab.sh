for i in 1 2; do
for cfg in "cur:" "q1:" "q1:RTOW_LEAF_VOTES=16" "q1:RTOW_LEAF_VOTES=40" "q1:RTOW_LEAF_VOTES=48"; do
  v=${cfg%%:*}; e=${cfg#*:}
  lib=$PWD/raytracing-one-weekend_amd/variants/$v.so; [ "$v" = cur ] && lib=$PWD/raytracing-one-weekend_amd/librtow.so
  line="$v [$e]"
  for sc in "suzanne" "mesh100k --spp 256"; do
    ms=$(env RTOW_LIB=$lib $e timeout -k 10 300 python scripts/bench_scene.py $sc --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['node_tests_per_segment'], d['prim_tests_per_segment'])") || exit 1
    line="$line | ${sc%% *} $ms"
  done
  echo "$line"
done; done
