cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_att; mkdir -p $O
python -m pytest tests/test_hip_modules.py -x -q 2>&1 | tail -5
for rep in 1 2; do for v in 0 1; do
echo "fuse $v: $(SLIDE_MODULE_FUSE_ATTEND=$v SLIDE_MODULE_PREC=fp16 python tools/time_decode.py 2>&1 | tail -1) | $(SLIDE_MODULE_FUSE_ATTEND=$v SLIDE_MODULE_PREC=fp16 python tools/time_encode.py 2>&1 | tail -1)"
done; done
