#!/bin/bash
# HBM-side bytes of the autoencoder decode leg (BASELINE configs[4], fp16 operands): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# passes over tools/time_decode.py (GPU box, via gpurun).  Summary: python tools/decode_hbm_summarize.py <tag> -> profiles/<tag>_decode_hbm_pmc.md
TAG=${1:-r05}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/dechbm_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
SLIDE_MODULE_PREC=fp16 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- python tools/time_decode.py > $OUT/fetch.log 2>&1
SLIDE_MODULE_PREC=fp16 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- python tools/time_decode.py > $OUT/write.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; tail -1 $OUT/fetch.log
