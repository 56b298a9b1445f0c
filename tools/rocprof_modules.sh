#!/bin/bash
# Module-level path profiles (GPU box, via gpurun): tools/rocprof_modules.sh <tag>
#   rocprofv3 --kernel-trace --stats of tools/time_decode.py (fp32 and fp16 operands) and tools/time_encode.py
# Summaries: python tools/rocprof_modules_summarize.py <tag>  -> profiles/<tag>_{decode,decode_fp16,encode}_kernel_stats.md
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profm_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode -o t -- python tools/time_decode.py > $OUT/decode.log 2>&1
SLIDE_MODULE_PREC=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_fp16 -o t -- python tools/time_decode.py > $OUT/decode_fp16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/encode -o t -- python tools/time_encode.py > $OUT/encode.log 2>&1
SLIDE_MODULE_PREC=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/encode_fp16 -o t -- python tools/time_encode.py > $OUT/encode_fp16.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/decode.log; tail -1 $OUT/decode_fp16.log; tail -1 $OUT/encode.log; tail -1 $OUT/encode_fp16.log
