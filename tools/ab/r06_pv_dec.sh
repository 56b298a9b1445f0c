# packed vectors on the module path: decode / encode parity tests, decode + encode rates with and without
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_pvd; mkdir -p $O
python -m pytest tests/test_hip_modules.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for v in 0 1; do
echo "pv $v: $(SLIDE_PACKED_VECS=$v SLIDE_MODULE_PREC=fp16 python tools/time_decode.py 2>&1 | tail -1) | $(SLIDE_PACKED_VECS=$v SLIDE_MODULE_PREC=fp16 python tools/time_encode.py 2>&1 | tail -1)"
done; done
