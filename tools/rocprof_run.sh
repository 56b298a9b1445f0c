#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of bench.py + separate PMC passes (HBM bytes) for one eager step.
# Outputs under gpurun_out/prof_$1 ; summaries are copied into profiles/ by tools/rocprof_summarize.py afterwards.
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python tools/profile_ops.py --reps 2 --batch 88 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python tools/profile_ops.py --reps 2 --batch 88 > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench.log | cut -c1-400
