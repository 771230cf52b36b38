#!/bin/bash
# PMC counter passes for the trace kernel (one rocprofv3 run per counter group; PMC is
# never combined with the trace domains gpurun refuses).  Usage: scripts/pmc_passes.sh TAG [bench args]
export TMPDIR=/tmp
TAG=${1:-x}; shift
mkdir -p gpurun_out
echo "nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null) affinity=$(python3 -c 'import os;print(len(os.sched_getaffinity(0)))')"
run() { name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-base --no-other-configs --no-end-to-end --no-reference-boundary $BENCH_ARGS > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 gpurun_out/pmc_${TAG}_$name.log; return 1; }; echo "pass $name ok"; }
BENCH_ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
