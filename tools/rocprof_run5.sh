#!/bin/bash
# Round-5 profile set (GPU box, via gpurun): tools/rocprof_run5.sh <tag>
#   trace        rocprofv3 --kernel-trace --stats of the DEFAULT bench arrangement (position DDPM split, feature DDPM fp16) in GRAPH replay
#   pmc_*        FETCH_SIZE / WRITE_SIZE / MFMA counters (separate passes) of one eager FEATURE step (88 samples per launch)
#   <tag>_pos    the same three passes over one eager POSITION step in the split arithmetic (512 samples: the arrangement's launch size)
# Summaries: python tools/rocprof_summarize.py <tag>; python tools/rocprof_summarize.py <tag>_pos 512 "--which pos --prec split"
TAG=${1:-r05a}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity --no-decode --replay graph > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python tools/profile_ops.py --reps 2 --batch 88 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python tools/profile_ops.py --reps 2 --batch 88 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY --output-format csv -d $OUT/pmc_mfma -o mfma -- python tools/profile_ops.py --reps 2 --batch 88 > $OUT/pmc_mfma.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv
OUTP=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_pos
rm -rf $OUTP; mkdir -p $OUTP/trace
cp $OUT/trace/*kernel_stats.csv $OUTP/trace/ 2>/dev/null
P="--which pos --prec split --batch 512 --reps 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUTP/pmc_fetch -o fetch -- python tools/profile_ops.py $P > $OUTP/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUTP/pmc_write -o write -- python tools/profile_ops.py $P > $OUTP/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY --output-format csv -d $OUTP/pmc_mfma -o mfma -- python tools/profile_ops.py $P > $OUTP/pmc_mfma.log 2>&1
find $OUT $OUTP -name "*agent_info.csv" -delete
du -sh $OUT $OUTP; tail -1 $OUT/bench.log | cut -c1-300; tail -3 $OUTP/pmc_mfma.log
