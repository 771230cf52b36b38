#!/bin/bash
# extra PMC passes: the vector-memory path (TA / TCP) — is a kernel that waits bound by L1 throughput or by latency?
# Two counters per pass (more per TA / TCP block "exceeds the capabilities of the hardware"), short timeouts (a
# refused counter set leaves rocprofv3 hanging after its error message).
# Usage: scripts/pmc_mem.sh TAG [bench args]
export TMPDIR=/tmp
TAG=${1:-x}; shift
BENCH_ARGS="$*"
run() { name=$1; shift; timeout -k 5 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-base --no-other-configs --no-end-to-end --no-reference-boundary $BENCH_ARGS > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { echo "pmc $name failed"; grep -m1 "error code" gpurun_out/pmc_${TAG}_$name.log; return 1; }; echo "pass $name ok"; }
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum &&
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
run tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum &&
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum &&
run tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum &&
run tcp4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum &&
run tcp5 TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum &&
run grbm GRBM_GUI_ACTIVE
