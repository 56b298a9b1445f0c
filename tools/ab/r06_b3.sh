# after a position-kernel change: engine parity tests (split + fp32 + chains), three bench runs, the position step's per-op times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_b3; mkdir -p $O
python -m pytest tests/test_hip_engine.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for rep in 1 2 3; do
echo "rep $rep: $(python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])")"
done
python tools/profile_ops.py --which pos --prec split --batch 512 2>&1 | grep "ATTN_TAIL\|total us"
python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-roofline --no-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(json.dumps(d['parity'])[:700])"
( export SLIDE_GX_DUAL=0 TL_PREC=split; python tools/ab/op_timeline.py pos 512 kind17 2>&1 | grep -v "amdgpu.ids\|no stamps" )
