#!/bin/bash
# Round-6 profile set (GPU box, via gpurun): tools/rocprof_run6.sh <tag>
#   tools/rocprof_run5.sh's passes under <tag> (kernel trace of the default bench arrangement; FETCH / WRITE / MFMA counters of one
#   eager FEATURE step at 88 samples per launch and of one POSITION step, split arithmetic, at 512), then the one-GPU BATCH CURVE
#   (VERDICT r5 item 5): the default arrangement at per-GPU batches 256 / 512 / 1024 / 2048 under rocprofv3 --kernel-trace --stats, and
#   the MFMA counters of one feature step at the sub-batch sizes those arrangements launch (88 / 176 / 344 / 688 samples per launch).
# Summaries: python tools/rocprof_summarize.py <tag>; python tools/rocprof_summarize.py <tag>_pos 512 "--which pos --prec split";
#            python tools/batch_curve_report.py <tag>  -> profiles/<tag>_batch_curve.md
TAG=${1:-r06a}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/rocprof_run5.sh $TAG
CUR=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_curve
rm -rf $CUR; mkdir -p $CUR
for b in 256 512 1024 2048; do
  python bench.py --gpus 1 --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $CUR/bench_$b.json 2> $CUR/bench_$b.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $CUR/trace_$b -o trace -- python bench.py --gpus 1 --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs --replay graph > $CUR/trace_$b.log 2>&1
  rm -f $CUR/trace_$b/*kernel_trace.csv
done
for n in 88 176 344 688; do
  python tools/profile_ops.py --reps 3 --batch $n > $CUR/ops_$n.txt 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY --output-format csv -d $CUR/mfma_$n -o mfma -- python tools/profile_ops.py --reps 2 --batch $n > $CUR/mfma_$n.log 2>&1
done
find $CUR -name "*agent_info.csv" -delete
du -sh $CUR; for b in 256 512 1024 2048; do cut -c1-160 $CUR/bench_$b.json; done
