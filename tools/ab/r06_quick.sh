# quick check of a kernel change: engine parity tests, three bench runs (--steps 300), one feature step's per-op times, timelines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_quick; mkdir -p $O
python -m pytest tests/test_hip_engine.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2 3; do
python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/b_$rep.json 2>$O/err_$rep.txt
echo "rep $rep: $(python -c "import json;d=json.load(open('$O/b_$rep.json'));print(d['value'], d['ms_per_step'])")"
done
python tools/profile_ops.py --batch 88 > $O/ops88.txt 2>&1; grep "total us" $O/ops88.txt
python tools/profile_ops.py --batch 688 > $O/ops688.txt 2>&1; grep "total us" $O/ops688.txt
export SLIDE_CHAIN_P=0 SLIDE_GX_DUAL=0; for k in $TLK; do python tools/ab/op_timeline.py feat 88 $k 2>&1 | grep -v amdgpu.ids; done
