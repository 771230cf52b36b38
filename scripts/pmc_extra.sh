#!/bin/bash
# extra PMC passes: LDS behaviour and the VALU instruction mix.  Usage: scripts/pmc_extra.sh TAG [bench args]
export TMPDIR=/tmp
TAG=${1:-x}; shift
BENCH_ARGS="$*"
run() { name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-base --no-other-configs --no-end-to-end --no-reference-boundary $BENCH_ARGS > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 gpurun_out/pmc_${TAG}_$name.log; return 1; }; echo "pass $name ok"; }
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES
run mix1 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32
run mix2 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU2 SQ_CYCLES SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INSTS_VALU
