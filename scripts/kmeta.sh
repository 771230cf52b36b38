#!/bin/bash
# Register / scratch / LDS metadata and instruction totals of the trace kernels of one arithmetic build, from a
# `hipcc -S` listing: scripts/kmeta.sh fast|strict|f32 [extra compiler flags]   (listing kept in /tmp/rtow_<build>.s)
set -e
B=${1:-fast}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="-ffp-contract=fast -fno-signed-zeros -fno-trapping-math -freciprocal-math -fno-math-errno"
[ "$B" = strict ] && FLAGS="-ffp-contract=off"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC $FLAGS "$@" -S --cuda-device-only \
  "$ROOT/raytracing-one-weekend_amd/csrc/rtow_trace_$B.hip" -o /tmp/rtow_$B.s 2>/dev/null
python3 - /tmp/rtow_$B.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', txt, re.S):
    blk = m.group(0)
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    short = re.sub(r'^_ZN4rtow12_GLOBAL__N_1\d+', '', name).replace('EEEvNS_11TraceParamsE', '')
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print(f"{short:44s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} scratch {g('private_segment_fixed_size'):>4} "
          f"spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')}")
PY
