# quick A/B of a tail-kernel change: bit-identity test, two alternating pairs against SLIDE_FM=0, per-op times at 688 / 88 samples
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_fm2; mkdir -p $O
python -m pytest tests/test_hip_engine.py -x -q -m gpu -k "fragment_major or matches_reference or pair_decomposition or chains_follow or golden" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for v in 0 1; do
SLIDE_FM=$v python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/fm_${v}_$rep.json 2>$O/err_${v}_$rep.txt
echo "fm $v rep $rep: $(python -c "import json;d=json.load(open('$O/fm_${v}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done
for b in 688 88; do python tools/profile_ops.py --batch $b > $O/ops$b.txt 2>&1; echo "== $b"; grep -h "ATTN_TAIL\|total us" $O/ops$b.txt; done
python tools/ab/op_timeline.py feat 688 kind16 2>&1 | grep -v amdgpu.ids; python tools/ab/op_timeline.py feat 88 kind16 2>&1 | grep -v amdgpu.ids
