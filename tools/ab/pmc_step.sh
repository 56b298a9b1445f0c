#!/bin/bash
# usage (on the GPU box): tools/ab/pmc_step.sh <tag>   -- SQ instruction counters of every kernel of one feature-denoiser step
TAG=$1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcstep_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/p -o p -- python tools/profile_ops.py --batch 128 --reps 1 > $OUT/p.log 2>&1
python tools/pmc_summarize.py $OUT/p "" > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
