# descriptor tables of the module path through pinned slots + asynchronous copies (SLIDE_PINNED_TABLES) against the synchronous upload
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_pin; mkdir -p $O
python -m pytest tests/test_hip_modules.py tests/test_hip_cli.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2 3; do for v in 0 1; do
echo "pinned $v: $(SLIDE_PINNED_TABLES=$v SLIDE_MODULE_PREC=fp16 python tools/time_decode.py 2>&1 | tail -1) | $(SLIDE_PINNED_TABLES=$v SLIDE_MODULE_PREC=fp16 python tools/time_encode.py 2>&1 | tail -1)"
done; done
