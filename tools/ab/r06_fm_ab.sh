# fragment-major u / mo (SLIDE_F_OUT_FM, attn_tail_rx_kernel<., true>) against chunk-major: bit-identity test, then three alternating pairs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_fm; mkdir -p $O
python -m pytest tests/test_hip_engine.py -x -q -m gpu -k "fragment_major or chunk_major or matches_reference or pair_decomposition" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for rep in 1 2 3; do for v in 0 1; do
SLIDE_FM=$v python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/fm_${v}_$rep.json 2>$O/err_${v}_$rep.txt
echo "fm $v rep $rep: $(python -c "import json;d=json.load(open('$O/fm_${v}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done
for v in 0 1; do
SLIDE_FM=$v python tools/profile_ops.py --batch 688 > $O/ops688_$v.txt 2>&1
SLIDE_FM=$v python tools/profile_ops.py --batch 88 > $O/ops88_$v.txt 2>&1
done
for f in ops688_0 ops688_1 ops88_0 ops88_1; do echo "== $f"; grep -h "ATTN_TAIL\|total us" $O/$f.txt; done
