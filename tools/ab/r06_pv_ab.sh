# packed epilogue vectors (SLIDE_EPI_PACKED_VECS: descriptors + vectors staged by LDS-DMA) against the pointer-chase staging
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_pv; mkdir -p $O
python -m pytest tests/test_hip_engine.py tests/test_hip_gemm.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2 3; do for v in 0 1; do
SLIDE_PACKED_VECS=$v python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/pv_${v}_$rep.json 2>$O/err_${v}_$rep.txt
echo "pv $v rep $rep: $(python -c "import json;d=json.load(open('$O/pv_${v}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done
for v in 0 1; do SLIDE_PACKED_VECS=$v python tools/profile_ops.py --batch 88 > $O/ops88_$v.txt 2>&1; grep "total us" $O/ops88_$v.txt; done
export SLIDE_CHAIN_P=0 SLIDE_GX_DUAL=0; for k in kind17 kind31; do python tools/ab/op_timeline.py feat 88 $k 2>&1 | grep -v amdgpu.ids; done
