#!/bin/bash
# Same-call A/B of library variants: scripts/ab_variants.sh OUT ROUNDS v1 v2 ...   (variants/NAME.so; "cur" = librtow.so)
# Per round and variant one line of scripts/scale_sweep.py (N = 1 at 100 / 500 spp, every rank's share at N = 8).
OUT=$1; R=$2; shift 2
for i in $(seq $R); do
  for v in "$@"; do
    lib=raytracing-one-weekend_amd/variants/$v.so
    [ "$v" = cur ] && lib=raytracing-one-weekend_amd/librtow.so
    echo -n "$v " >> $OUT
    timeout -k 10 300 python scripts/scale_sweep.py --lib $lib "" 2>/dev/null | grep defaults >> $OUT || exit 1
  done
done
