#!/bin/bash
# Round-3 profile set (GPU box, via gpurun): tools/rocprof_run3.sh <tag> [batch of the PMC passes, default 88 = a bench sub-batch]
#   trace      rocprofv3 --kernel-trace --stats of the bench command in GRAPH replay (eager replay is host-bound under the profiler)
#   pmc_fetch / pmc_write   HBM-side bytes (separate passes) of one eager feature step
#   pmc_mfma   SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_WAIT_ANY per kernel
# Summaries: python tools/rocprof_summarize.py <tag>  (-> profiles/<tag>_kernel_stats.md, _hbm_pmc.md, _mfma_pmc.md)
TAG=${1:-r03a}
PB=${2:-88}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity --no-decode --replay graph > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python tools/profile_ops.py --reps 2 --batch $PB > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python tools/profile_ops.py --reps 2 --batch $PB > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY --output-format csv -d $OUT/pmc_mfma -o mfma -- python tools/profile_ops.py --reps 2 --batch $PB > $OUT/pmc_mfma.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv
tail -1 $OUT/bench.log | cut -c1-300
