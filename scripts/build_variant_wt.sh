#!/bin/bash
# Build librtow.so of the WORKING TREE with extra compiler flags into raytracing-one-weekend_amd/variants/NAME.so
# (same-device A/B runs: RTOW_LIB=raytracing-one-weekend_amd/variants/NAME.so python bench.py ...):
#   scripts/build_variant_wt.sh NAME "-DRTOW_SOME_EXPERIMENT=1 ..."
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/raytracing-one-weekend_amd"
cp -r "$ROOT/include" "$TMP/"
cp -r "$ROOT/raytracing-one-weekend_amd/csrc" "$ROOT/raytracing-one-weekend_amd/host" "$ROOT/raytracing-one-weekend_amd/Makefile" "$TMP/raytracing-one-weekend_amd/"
make -C "$TMP/raytracing-one-weekend_amd" -j8 librtow.so EXTRA="$FLAGS" > "$TMP/build.log" 2>&1 || { tail -30 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/raytracing-one-weekend_amd/variants"
cp "$TMP/raytracing-one-weekend_amd/librtow.so" "$ROOT/raytracing-one-weekend_amd/variants/$NAME.so"
rm -rf "$TMP"
echo "built variants/$NAME.so from the working tree with EXTRA='$FLAGS'"
