#!/bin/bash
# Build librtow.so of another git revision into raytracing-one-weekend_amd/variants/NAME.so
# (for same-device A/B runs: RTOW_LIB=raytracing-one-weekend_amd/variants/NAME.so python bench.py ...)
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" raytracing-one-weekend_amd include | tar -x -C "$TMP"
make -C "$TMP/raytracing-one-weekend_amd" -j8 librtow.so > /dev/null
mkdir -p "$ROOT/raytracing-one-weekend_amd/variants"
cp "$TMP/raytracing-one-weekend_amd/librtow.so" "$ROOT/raytracing-one-weekend_amd/variants/$NAME.so"
rm -rf "$TMP"
echo "built variants/$NAME.so from $REV"
