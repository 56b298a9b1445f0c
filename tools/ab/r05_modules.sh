cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05m; O=gpurun_out/r05m
timeout 1200 python -m pytest tests/test_hip_modules.py tests/test_hip_cli.py -q -m gpu -x > $O/modtest.log 2>&1; tail -6 $O/modtest.log
for v in 1 0; do
SLIDE_MODULE_SPLIT_QK=$v SLIDE_MODULE_PREC=fp16 python tools/time_decode.py 2>&1 | tail -1
SLIDE_MODULE_SPLIT_QK=$v SLIDE_MODULE_PREC=fp16 python tools/time_encode.py 2>&1 | tail -1
done
SLIDE_MODULE_SPLIT_QK=1 SLIDE_MODULE_PREC=fp16 python tools/time_decode.py 2>&1 | tail -1
SLIDE_MODULE_SPLIT_QK=0 SLIDE_MODULE_PREC=fp16 python tools/time_decode.py 2>&1 | tail -1
