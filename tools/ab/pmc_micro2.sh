#!/bin/bash
# usage (on the GPU box): tools/ab/pmc_micro2.sh <tag> <shape> [<shape>...]   -- LDS / MFMA busy counters of single GEMM launches
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/p3 -o p3 -- python tools/ab/gemm_micro.py "$@" > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES --output-format csv -d $OUT/p4 -o p4 -- python tools/ab/gemm_micro.py "$@" > $OUT/p4.log 2>&1
tail -2 $OUT/p3.log
