export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profm_r05
rm -rf $OUT; mkdir -p $OUT
SLIDE_MODULE_PREC=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_fp16 -o t -- python tools/time_decode.py > $OUT/decode_fp16.log 2>&1
SLIDE_MODULE_PREC=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/encode_fp16 -o t -- python tools/time_encode.py > $OUT/encode_fp16.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/decode_fp16.log; tail -1 $OUT/encode_fp16.log
bash tools/decode_hbm.sh r05
