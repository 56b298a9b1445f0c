cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_split; mkdir -p $O
python -m pytest tests/test_hip_engine.py -x -q -k "not full_chains and not optin" 2>&1 | tail -3
for rep in 1 2 3; do for v in 0 1; do
SLIDE_SA_SPLIT=$v python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs > $O/s_${v}_$rep.json 2>/dev/null
echo "split $v rep $rep: $(python -c "import json;d=json.load(open('$O/s_${v}_$rep.json'));print(d['value'], d['ms_per_step'])")"
done; done
