#!/bin/bash
# One line per environment setting for the mesh workloads: scripts/knob_scenes.sh OUT "ENV1=a" "ENV2=b ENV3=c" ...
# (kernel ms / Msamples/s of suzanne and of the big mesh at 256 spp; "" = the defaults)
OUT=$1; shift
for e in "$@"; do
  line="[$e]"
  for sc in "suzanne" "mesh100k --spp 256"; do
    ms=$(env $e timeout -k 10 300 python scripts/bench_scene.py $sc --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['Msamples_per_s'])") || exit 1
    line="$line | ${sc%% *} $ms"
  done
  echo "$line" >> $OUT
done
