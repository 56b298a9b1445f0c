#!/bin/bash
# Training-step profile (GPU box, via gpurun): tools/rocprof_train.sh <tag> [batch]
#   rocprofv3 --kernel-trace --stats of tools/time_train.py <batch> --eager-only (the eager step: the graphed step replays the same kernels)
# Summary: python tools/rocprof_modules_summarize.py <tag> train<batch>  (profiles/<tag>_train<batch>_kernel_stats.md)
TAG=${1:-r04}; B=${2:-32}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profm_$TAG
mkdir -p $OUT; rm -rf $OUT/train$B
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train$B -o t -- python tools/time_train.py $B --eager-only > $OUT/train$B.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/train$B.log
