#!/bin/bash
# usage (on the GPU box): tools/ab/pmc_icache.sh <tag> <shape> [<shape>...]   -- instruction-fetch counters of single GEMM launches
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmci_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -40 > $OUT/avail.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/p1 -o p1 -- python tools/ab/gemm_micro.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES --output-format csv -d $OUT/p2 -o p2 -- python tools/ab/gemm_micro.py "$@" > $OUT/p2.log 2>&1
python tools/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/avail.txt | head -30; tail -5 $OUT/p1.log; tail -3 $OUT/p2.log; cat $OUT/summary.txt
