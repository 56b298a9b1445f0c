#!/bin/bash
# usage (on the GPU box): tools/ab/pmc_micro.sh <tag> <shape> [<shape>...]   -- SQ counters of single GEMM launches
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p1 -o p1 -- python tools/ab/gemm_micro.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -o p2 -- python tools/ab/gemm_micro.py "$@" > $OUT/p2.log 2>&1
find $OUT -name "*counter_collection.csv"
tail -3 $OUT/p1.log
